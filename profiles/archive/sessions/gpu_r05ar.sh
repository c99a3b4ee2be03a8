#!/bin/bash
# session r05ar: the LDS allocation granule (how many blocks does a CU hold at 31.5 ... 33.5 KB per block?), then the round's evidence on the final sources
OUT=gpurun_out/r05ar; mkdir -p $OUT; export TMPDIR=/tmp
for kt in 186 188 190 192 194 196 198 200; do
python tools/timeline.py --kt $kt --cfg 24 2>>$OUT/err.txt | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(json.dumps({'kt': d['audio_per_tile'], 'lds': d['lds_bytes'], 'max_resident': d['max_resident_blocks_last_cu'], 'resident_mean': round(d['resident_blocks_per_cu']['mean'], 2), 'ms': d['ms_per_call_without']}))
" | tee -a $OUT/granule.jsonl
done
bash scripts/gpu_round.sh r05
