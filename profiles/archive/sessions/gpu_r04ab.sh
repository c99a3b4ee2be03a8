#!/bin/bash
# session r04ab: fmd_demod_check behind one stream synchronisation (page-locked report head): boundary / guard tests + check_per_step before / after
OUT=gpurun_out/r04ab; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_f64_guard.py tests/test_gpu_sink.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
for i in 1 2; do
  for v in base new; do
    L=libfmd_hip.so; [ $v = base ] && L=libfmd_hip_base.so
    FMD_LIB=$PWD/rtl-sdr-rs_amd/$L python3 bench.py --no-cpu 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'headline', b['ms_per_step'], 'check_per_step', b['extra']['check_per_step']['ms_per_step'])"
  done
done | tee $OUT/check.txt
