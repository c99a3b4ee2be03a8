#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, a bench line, and a rocprofv3 kernel-trace summary.
# Usage: scripts/gpu_check.sh <tag> [pytest-args...]
TAG=${1:-r01}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocminfo" ; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4
nproc; free -g | head -2
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -x -q -m gpu "$@" 2>&1 | tail -40 | tee $OUT/pytest_gpu.log
echo "== bench"
timeout 600 python bench.py 2>$OUT/bench.err | tee $OUT/bench.json
tail -5 $OUT/bench.err
echo "== rocprofv3 kernel-trace"
timeout 600 rocprofv3 --kernel-trace --stats -S -u usec -d $OUT/prof -o trace -f csv -- python3 bench.py --no-cpu > $OUT/prof_bench.json 2> $OUT/prof.err
tail -30 $OUT/prof.err
ls -R $OUT/prof | head -20
