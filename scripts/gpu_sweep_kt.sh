#!/bin/bash
# A/B sweep of the tiling inside one box: interleaved rounds, prints ms_per_step per kt.
for round in 1 2; do
for kt in "$@"; do
  python bench.py --steps 200 --warmup 20 --no-cpu --no-extra --kt $kt 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print('kt=%4d round=$round ms=%.4f frac=%.3f lds=%d' % ($kt, r['ms_per_step'], r['roofline']['frac'], r['config']['tiling']['lds_bytes']))"
done
done
