#!/bin/bash
# session r04c: two-rank segfault with faulthandler; register-form fused FIR kernel: parity + A/B; scratch-free A/B vs the r03 build
OUT=gpurun_out/r04c; mkdir -p $OUT; export TMPDIR=/tmp
echo "== two-rank line (faulthandler)"
PYTHONFAULTHANDLER=1 timeout 600 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --settle 10 --channels 1024 --min-timed-s 0.05 --cpu-seconds 1 --power-only > $OUT/two_rank.json 2> $OUT/two_rank.err; echo "rc=$?"
tail -40 $OUT/two_rank.err | cut -c1-200
cut -c1-200 $OUT/two_rank.json
echo "== firdemod tests"
timeout 1500 python -m pytest tests/test_firdemod.py -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_firdemod.log
echo "== firdemod A/B (experiment library)"
export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
for r in 1 2; do for ng in 0 4 5 6; do FMD_FD_REG=$ng python tools/bench_firdemod.py 2>/dev/null | cut -c1-330; done; done | tee $OUT/fd_ab.jsonl
unset FMD_LIB
echo "== demod A/B: r03 build vs now (scratch-free f64 callee, named v_cvt)"
bash scripts/gpu_ablibs.sh "--cfg ref --cfg 24 --cfg 4,256000,48000 --cfg 7,166666,32000" r03=libfmd_hip_r03.so new=libfmd_hip.so 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_scratch.txt
bash scripts/gpu_ablibs.sh "--cfg ref --cfg 24" r03=libfmd_hip_r03.so new=libfmd_hip.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_scratch.txt
echo "== full suite"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_gpu.log
