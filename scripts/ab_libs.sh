#!/bin/bash
# Alternating-process A/B of library builds on one box.  Usage: scripts/ab_libs.sh <outfile> <rounds> "<ab.py args>" name=path ...
# (name=  with an empty path is the shipped library)
OUT=$1; R=$2; ARGS=$3; shift 3
for r in $(seq 1 $R); do
  for v in "$@"; do
    name=${v%%=*}; path=${v#*=}
    if [ -n "$path" ]; then export FMD_LIB=$PWD/$path; else unset FMD_LIB; fi
    python tools/ab.py --rounds 2 $ARGS $name: 2>/dev/null | grep '^{"cfg"' >> $OUT
  done
done
unset FMD_LIB
python3 - $OUT <<'PY'
import json, sys, collections
acc = collections.OrderedDict()
for l in open(sys.argv[1]):
    d = json.loads(l); acc.setdefault((tuple(d["cfg"]), d["variant"]), []).extend(d["ms"])
base = {}
for (cfg, v), ms in acc.items():
    ms = sorted(ms); med = ms[len(ms) // 2]
    base.setdefault(cfg, med)
    print("%-22s %-8s median %.4f  (%+.1f %%)  min %.4f max %.4f n=%d" % (cfg, v, med, 100 * (med / base[cfg] - 1), ms[0], ms[-1], len(ms)))
PY
