#!/bin/bash
# session r04j: stand-alone FIR kernel, round-3 prologue vs hot block (interleaved); FIR tests
OUT=gpurun_out/r04j; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fir.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3 4; do
  for v in firold fira new; do
    if [ $v = new ]; then unset FMD_LIB; else export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_$v.so; fi
    echo "$v $(python tools/bench_fir.py 2>/dev/null | python3 -c 'import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["ms_per_call"], r.get("hbm_frac_of_8TBps"))')"
  done
done | tee $OUT/fir_ab.txt
unset FMD_LIB
