#!/bin/bash
# session r04r: the f32 discriminator's quotient by nearest-integer + one clamped fma (3 instructions fewer); the complex
# product by fma(.., 0) instead of mul + (+ 0.0): parity + A/B
OUT=gpurun_out/r04r; mkdir -p $OUT; export TMPDIR=/tmp
FMD_FUZZ_CASES=100 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_variants.py tests/test_e2e_digests.py tests/test_firdemod.py tests/test_gpu_f64_guard.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_cmulfma.so FMD_FUZZ_CASES=100 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_e2e_digests.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest_cmulfma.log
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg ref --cfg 24 --cfg 12,192000,32000 --cfg 8,250000,44100 --cfg 7,170000,32000 --cfg 5,250000,44100 --cfg 4,300000,50000 --cfg 2,500000,32000 --cfg 1,48000,48000" base=libfmd_hip_base.so new=libfmd_hip.so cmulfma=libfmd_hip_cmulfma.so 2>&1 | grep '^{"cfg"'; done > $OUT/ab_disc.txt
python3 tools/ab_summary.py $OUT/ab_disc.txt
python3 tools/bench_firdemod.py 2>/dev/null | tail -1 | cut -c1-400
