#!/usr/bin/env python3
"""Mean / min / max per (configuration, variant) of the JSON lines tools/ab.py (scripts/gpu_ablibs.sh) printed; the last
column is each variant's mean relative to the first variant of its configuration.  Usage: tools/ab_summary.py file..."""
import collections, json, sys
acc = collections.OrderedDict()
for f in sys.argv[1:]:
    for l in open(f):
        if not l.startswith('{"cfg"'):
            continue
        try:
            r = json.loads(l)
        except ValueError:                      # a line the session script cut short: take the fields that are complete
            import re
            m = re.match(r'\{"cfg": \[(\d+), (\d+), (\d+)\], "variant": "(\w+)", "ms": \[([\d., ]+)\]', l)
            if not m:
                continue
            r = {"cfg": [int(m.group(1)), int(m.group(2)), int(m.group(3))], "variant": m.group(4), "ms": [float(x) for x in m.group(5).split(",")]}
        acc.setdefault(tuple(r["cfg"]), collections.OrderedDict()).setdefault(r["variant"], []).extend(r["ms"])
for cfg, vs in acc.items():
    base = None
    for v, ms in vs.items():
        mean = sum(ms) / len(ms)
        base = base or mean
        print("D=%-3d %7d->%-6d %-8s n=%d mean %.4f  min %.4f max %.4f  %+.2f %%" % (cfg[0], cfg[1], cfg[2], v, len(ms), mean, min(ms), max(ms), 100 * (mean / base - 1)))
