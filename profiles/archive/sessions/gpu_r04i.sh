#!/bin/bash
# session r04i: biased accumulators instead of int -> f32 conversions (A/B + parity); the stand-alone FIR kernel's hot-block prologue
OUT=gpurun_out/r04i; mkdir -p $OUT; export TMPDIR=/tmp
echo "== parity"
FMD_FUZZ_CASES=120 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_fir.py tests/test_gpu_variants.py tests/test_e2e_digests.py -x -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest.log
echo "== bias A/B"
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg ref --cfg 24 --cfg 4,256000,48000 --cfg 7,166666,32000 --cfg 5,250000,44100 --cfg 2,500000,32000 --cfg 8,250000,44100 --cfg 12,192000,32000" nobias=libfmd_hip_nobias.so new=libfmd_hip.so 2>&1 | grep -v amdgpu.ids | cut -c1-110; done | tee $OUT/ab_bias.txt
echo "== FIR"
for i in 1 2 3; do python tools/bench_fir.py 2>/dev/null | cut -c1-200; done | tee $OUT/fir.txt
