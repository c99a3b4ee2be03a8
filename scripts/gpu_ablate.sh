#!/bin/bash
# Ablation sweep with the -DFMD_EXPERIMENT library (tuning only).
export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
for round in 1 2; do
for dbg in "$@"; do
  FMD_DBG=$dbg python bench.py --steps 200 --warmup 20 --no-cpu 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print('dbg=%2d round=$round ms=%.4f frac=%.3f' % ($dbg, r['ms_per_step'], r['roofline']['frac']))"
done
done
