#!/bin/bash
# session r06ag: cache policy of the AUDIO STORES (global_store_short: default against sc1, sc0 sc1, sc1 nt, sc0; variant builds), same
# process; with a parity run on one variant (the policy must not change a byte)
OUT=gpurun_out/r06ag; mkdir -p $OUT; export TMPDIR=/tmp
L=$PWD/rtl-sdr-rs_amd
FMD_LIB=$L/libfmd_hip_st2.so timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "configs_batched_random or config1" > $OUT/parity_st2.log 2>&1; tail -1 $OUT/parity_st2.log
timeout 1500 python tools/ab_libs.py --rounds 4 --cfg 24 --cfg ref --cfg 5,250000,44100 --cfg 4,256000,48000 default=$L/libfmd_hip_r06b.so sc1=$L/libfmd_hip_st1.so sc0sc1=$L/libfmd_hip_st2.so sc1nt=$L/libfmd_hip_st3.so sc0=$L/libfmd_hip_st4.so 2>/dev/null | tee $OUT/ab.txt | cut -c1-200
