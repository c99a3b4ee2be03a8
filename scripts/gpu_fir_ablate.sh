#!/bin/bash
# FIR ablation sweep with the -DFMD_EXPERIMENT library (tuning only): bit 0 no MFMA, 1 no loads, 2 no stores, 3 no history kernel.
export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
for round in 1 2; do
for dbg in "$@"; do
  FMD_DBG=$dbg python tools/bench_fir.py 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print('dbg=%2d round=$round ms=%.4f frac=%.3f' % ($dbg, r['ms_per_call'], r['hbm_frac_of_8TBps']))"
done
done
