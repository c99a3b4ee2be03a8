#!/bin/bash
# session r04k: fused register-form kernel, tap fragments behind (fdold) vs in front of the DMAs (new)
OUT=gpurun_out/r04k; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_firdemod.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3 4; do
  for v in fdold new; do
    if [ $v = new ]; then unset FMD_LIB; else export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_$v.so; fi
    echo "$v $(python tools/bench_firdemod.py 2>/dev/null | python3 -c 'import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["ms"], r["ms_all"], r["frac"], r["kernel"])')"
  done
done | tee $OUT/fd_ab.txt
unset FMD_LIB
