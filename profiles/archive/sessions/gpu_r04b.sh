#!/bin/bash
# session r04b: suite on the refactored sources, A/B r03 / old resampler / new, region PMC, full bench line
OUT=gpurun_out/r04b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_gpu.log
bash scripts/gpu_ablibs.sh "--cfg ref --cfg 24 --cfg 4,256000,48000 --cfg 7,166666,32000" r03=libfmd_hip_r03.so rs0=libfmd_hip_rs0.so new=libfmd_hip.so 2>&1 | tee $OUT/ab_libs.txt
bash scripts/gpu_pmc_regions.sh r04b > $OUT/pmc_regions.log 2>&1
timeout 900 python bench.py 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-300
tail -3 $OUT/bench.err
