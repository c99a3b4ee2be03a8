#!/bin/bash
# session r04v: which of the two fused-kernel changes of r04u costs the 1.2 %?  base / both / without C-operand constants / without bit masks
OUT=gpurun_out/r04v; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2 3; do
  for v in base hip nocinit nomasks; do
    L=libfmd_hip_$v.so; [ $v = hip ] && L=libfmd_hip.so
    echo "$v $(FMD_LIB=$PWD/rtl-sdr-rs_amd/$L python3 tools/bench_firdemod.py 2>/dev/null | tail -1 | cut -c60-130)"
  done
done | tee $OUT/ab_fused.txt
