#!/bin/bash
# session r04d: two-rank line x5; register-form kernel tests, NG sweep, PMC; builtin vs named conversion A/B
OUT=gpurun_out/r04d; mkdir -p $OUT; export TMPDIR=/tmp
echo "== two-rank line x5"
for i in 1 2 3 4 5; do
  PYTHONFAULTHANDLER=1 timeout 600 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --settle 10 --channels 1024 --min-timed-s 0.05 --cpu-seconds 1 --power-only > $OUT/two_rank_$i.json 2> $OUT/two_rank_$i.err
  echo "run $i rc=$? line=$(wc -c < $OUT/two_rank_$i.json)"; grep -v "amdgpu.ids\|c10d\|Gloo" $OUT/two_rank_$i.err | tail -25 | cut -c1-200
done
echo "== firdemod tests"
timeout 1500 python -m pytest tests/test_firdemod.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest_firdemod.log
echo "== firdemod A/B (experiment library)"
export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
for r in 1 2; do
  FMD_FD_REG=0 python tools/bench_firdemod.py 2>/dev/null | cut -c1-330
  FMD_FD_REG=5 python tools/bench_firdemod.py 2>/dev/null | cut -c1-330
  FMD_FD_REG=6 FMD_FD_LDS=23400 python tools/bench_firdemod.py 2>/dev/null | cut -c1-330
  FMD_FD_REG=5 FMD_FD_KT=18 python tools/bench_firdemod.py 2>/dev/null | cut -c1-330
done | tee $OUT/fd_ab.jsonl
unset FMD_LIB
echo "== PMC register form"
bash scripts/gpu_pmc_firdemod.sh r04d_pmc_fd > $OUT/pmc_fd.log 2>&1; tail -45 $OUT/pmc_fd.log | head -40
echo "== cvt A/B"
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg 4,256000,48000 --cfg ref --cfg 24 --cfg 7,166666,32000 --rounds 1" cvtb=libfmd_hip_cvtb.so new=libfmd_hip.so 2>&1 | grep -v amdgpu.ids | cut -c1-110; done | tee $OUT/ab_cvt.txt
