#!/bin/bash
# session r05m: tile sizes on the new kernels across the domain (experiment build, FMD_KT), 3 rounds each, same process per configuration
OUT=gpurun_out/r05m; mkdir -p $OUT; export TMPDIR=/tmp
X=rtl-sdr-rs_amd/libfmd_hip_exp.so
sweep() { # cfg  kt...
  cfg=$1; shift
  V="base=$X"; for k in "$@"; do V="$V k$k=$X@FMD_KT=$k"; done
  python tools/ab_libs.py --rounds 3 --cfg $cfg $V 2>/dev/null >> $OUT/kt.jsonl
}
: > $OUT/kt.jsonl
sweep 24 68 76 84 88 92 96 100 102
sweep ref 128 144 160 176 192 208 224 240 248
sweep 12,192000,32000 70 80 90 100 106 112 118 122 130
sweep 16,150000,32000 54 64 74 84 94 100 104 112 118
sweep 64,37500,8000 14 18 20 22 24 27 30 34
sweep 8,250000,44100 100 120 140 160 180 200 220 240
sweep 5,250000,44100 128 160 192 208 224 240 256 272
sweep 7,166666,32000 128 160 176 192 208 224 256
sweep 14,224000,32000 48 56 64 80 88 96 104
python3 - $OUT/kt.jsonl <<'PY'
import json, re, sys
for l in open(sys.argv[1]):
    d = json.loads(l); m = re.search(r'kt=(\d+) lds=(\d+) tiles=(\d+)', d['kernel'])
    print('%3d %-6s kt=%-4s lds=%-6s tiles=%-3s %.4f %+.2f' % (d['cfg'][0], d['build'], m.group(1), m.group(2), m.group(3), d['median_ms'], d['vs_first_pct']))
PY
