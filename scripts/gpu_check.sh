#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, a bench line, and a rocprofv3 kernel-trace summary.
# Usage: scripts/gpu_check.sh <tag> [pytest-args...]      exit code = pytest's (bench / profile failures: 10 / 11)
set -o pipefail
TAG=${1:-r02}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocminfo" ; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4
nproc; free -g | head -2
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -x -q -m gpu "$@" 2>&1 | tail -40 | tee $OUT/pytest_gpu.log
RC=${PIPESTATUS[0]}
echo "pytest rc=$RC"
echo "== bench"
timeout 900 python bench.py 2>$OUT/bench.err | tee $OUT/bench.json || RC2=10
tail -5 $OUT/bench.err
echo "== rocprofv3 kernel-trace"
timeout 900 rocprofv3 --kernel-trace --stats -S -u usec -d $OUT/prof -o trace -f csv -- python3 bench.py --no-cpu --no-extra > $OUT/prof_bench.json 2> $OUT/prof.err || RC3=11
tail -30 $OUT/prof.err
ls -R $OUT/prof | head -20
[ "$RC" != "0" ] && exit $RC
exit ${RC2:-${RC3:-0}}
