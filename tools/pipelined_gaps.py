#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace of tools/pipelined_trace.py: per cadence (300 launches each, in order) the median kernel duration and
the median gap from one kernel's end to the next one's start.  Usage: tools/pipelined_gaps.py <dir with *kernel_trace.csv>"""
import csv, glob, json, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if "fmd_demod" in r["Kernel_Name"]]
rows.sort()
out = {}
med = lambda v: sorted(v)[len(v) // 2]
for k, name in enumerate(("warm-up", "bare", "check_behind_2", "check_prev", "check")):
    seg = rows[300 * k:300 * (k + 1)]
    if len(seg) < 300:
        break
    dur = [e - s for s, e in seg]
    gap = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
    big = [g for g in gap if g > 20000]
    out[name] = {"kernel_us_median": round(med(dur) / 1e3, 2), "kernel_us_mean": round(sum(dur) / len(dur) / 1e3, 2),
                 "gap_us_median": round(med(gap) / 1e3, 2), "gap_us_p90": round(sorted(gap)[int(0.9 * len(gap))] / 1e3, 2),
                 "gap_us_mean": round(sum(gap) / len(gap) / 1e3, 2), "gap_us_max": round(max(gap) / 1e3, 2),
                 "gaps_over_20us": len(big), "their_sum_us": round(sum(big) / 1e3, 1),
                 "period_us_median": round(med([seg[i + 1][0] - seg[i][0] for i in range(len(seg) - 1)]) / 1e3, 2),
                 "span_us_per_launch": round((seg[-1][1] - seg[0][0]) / len(seg) / 1e3, 2),
                 "first_60_kernel_us": [round(x / 1e3) for x in dur[:60]], "first_60_gap_us": [round(x / 1e3) for x in gap[:60]]}
print(json.dumps(out))
