#!/bin/bash
# session r04f: register-form kernel after the prologue rework (tests, NG sweep, PMC); packed-fma complex product A/B
OUT=gpurun_out/r04f; mkdir -p $OUT; export TMPDIR=/tmp
echo "== firdemod tests"
timeout 1500 python -m pytest tests/test_firdemod.py -x -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest_firdemod.log
echo "== firdemod NG sweep (experiment library)"
export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
for r in 1 2 3; do
  for ng in 0 5 6 7 8; do FMD_FD_REG=$ng python tools/bench_firdemod.py 2>/dev/null | cut -c1-300; done
done | tee $OUT/fd_ab.jsonl
unset FMD_LIB
echo "== PMC register form (shipped library)"
bash scripts/gpu_pmc_firdemod.sh r04f_pmc_fd > $OUT/pmc_fd.log 2>&1; grep -E "SQ_INSTS|SQ_WAVES|WAVE_CYCLES|BANK|IDX_ACTIVE|FETCH|WRITE" $OUT/pmc_fd.log
echo "== packed-fma complex product A/B"
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg 4,256000,48000 --cfg ref --cfg 24 --cfg 2,500000,32000 --cfg 8,250000,44100" pkc=libfmd_hip_pkc.so new=libfmd_hip.so 2>&1 | grep -v amdgpu.ids | cut -c1-110; done | tee $OUT/ab_pkc.txt
echo "== fuzz on the pkc build"
FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_pkc.so FMD_FUZZ_CASES=150 timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or silence or axis or configs_batched" 2>&1 | tail -4
