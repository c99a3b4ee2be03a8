#!/bin/bash
# session r06af: cache policy of the LDS-DMA staging loads (aux of global_load_lds: sc0 = 1, nt = 2, sc1 = 16) -- the shipped nt against
# default, sc0, sc0 nt, sc1, sc1 nt, sc0 sc1 nt (variant builds), same process; time and, separately, nothing else
OUT=gpurun_out/r06af; mkdir -p $OUT; export TMPDIR=/tmp
L=$PWD/rtl-sdr-rs_amd
timeout 1500 python tools/ab_libs.py --rounds 4 --cfg 24 --cfg ref --cfg 5,250000,44100 --cfg 8,250000,44100 nt=$L/libfmd_hip_r06b.so default=$L/libfmd_hip_aux0.so sc0=$L/libfmd_hip_aux1.so sc0nt=$L/libfmd_hip_aux3.so sc1=$L/libfmd_hip_aux16.so sc1nt=$L/libfmd_hip_aux18.so sc0sc1nt=$L/libfmd_hip_aux19.so 2>/dev/null | tee $OUT/ab.txt | cut -c1-200
