#!/bin/bash
# session r04g: larger columns for the register-form kernel; the boxcar kernel with no early exit / one scalar round trip in front of its DMAs
OUT=gpurun_out/r04g; mkdir -p $OUT; export TMPDIR=/tmp
echo "== parity (changed prologues)"
timeout 1500 python -m pytest tests/test_firdemod.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_variants.py -x -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest.log
echo "== firdemod NG sweep (experiment library)"
export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
for r in 1 2; do
  for ng in 5 8 10 12; do FMD_FD_REG=$ng python tools/bench_firdemod.py 2>/dev/null | cut -c1-300; done
done | tee $OUT/fd_ab.jsonl
unset FMD_LIB
echo "== boxcar prologue A/B"
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg ref --cfg 24 --cfg 4,256000,48000 --cfg 7,166666,32000 --cfg 16,150000,32000 --cfg 64,37500,8000" base=libfmd_hip_base.so new=libfmd_hip.so 2>&1 | grep -v amdgpu.ids | cut -c1-110; done | tee $OUT/ab_prologue.txt
