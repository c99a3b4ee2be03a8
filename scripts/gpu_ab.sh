#!/bin/bash
# Interleaved A/B of environment-selected variants inside one box.  Usage: gpu_ab.sh "VAR=val ..." "VAR=val ..." ...
for round in 1 2 3; do
for v in "$@"; do
  env $v python bench.py --steps 200 --warmup 20 --no-cpu 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print('%-40s round=$round ms=%.4f frac=%.3f' % ('$v', r['ms_per_step'], r['roofline']['frac']))"
done
done
