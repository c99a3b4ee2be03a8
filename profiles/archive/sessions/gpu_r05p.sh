#!/bin/bash
# session r05p: where is the threshold of the second plan?  rows with a vector share of 0.86 - 0.93 on 17 KB tiles; downsample 15 / 32 tile sizes
OUT=gpurun_out/r05p; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
sweep() { cfg=$1; shift; V="base=$X"; for k in "$@"; do V="$V k$k=$X@FMD_KT=$k"; done; python tools/ab_libs.py --rounds 3 --cfg $cfg $V 2>/dev/null >> $OUT/kt.jsonl; }
: > $OUT/kt.jsonl
sweep 10,250000,48000 130 138 144 150 158
sweep 9,180000,32000 140 150 158 166
sweep 8,250000,44100 150 158 166 172
sweep 15,240000,32000 52 56 60 68 72 76
sweep 32,512000,32000 14 16 17 18
python3 - $OUT/kt.jsonl <<'PY'
import json, re, sys
for l in open(sys.argv[1]):
    d = json.loads(l); m = re.search(r'kt=(\d+) lds=(\d+) tiles=(\d+)', d['kernel'])
    print('%3d %-6s kt=%-4s lds=%-6s tiles=%-3s %.4f %+.2f' % (d['cfg'][0], d['build'], m.group(1), m.group(2), m.group(3), d['median_ms'], d['vs_first_pct']))
PY
