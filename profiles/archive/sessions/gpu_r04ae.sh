#!/bin/bash
# session r04ae: compiler scheduling strategies (-mllvm -amdgpu-sched-strategy=max-ilp / max-memory-clause) and -O2 against the shipped -O3 build
OUT=gpurun_out/r04ae; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg ref --cfg 24 --cfg 5,250000,44100 --cfg 2,500000,32000 --cfg 8,250000,44100" base=libfmd_hip_base.so maxilp=libfmd_hip_maxilp.so memclause=libfmd_hip_memclause.so o2=libfmd_hip_o2.so 2>&1 | grep '^{"cfg"'; done > $OUT/ab_sched.txt
python3 tools/ab_summary.py $OUT/ab_sched.txt
for v in base maxilp memclause o2; do echo "$v $(FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_$v.so python3 tools/bench_firdemod.py 2>/dev/null | tail -1 | cut -c60-130)"; done
