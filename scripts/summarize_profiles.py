#!/usr/bin/env python3
"""Turn the raw outputs of scripts/gpu_check.sh <tag> and scripts/gpu_pmc.sh <tag>_pmc (under gpurun_out/) into the
committed summaries under profiles/.  Usage: summarize_profiles.py <tag> [round-prefix, default r01]"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pmc, out = os.path.join(root, "gpurun_out", tag), os.path.join(root, "gpurun_out", tag + "_pmc"), os.path.join(root, "profiles")
KERNEL = "fmd_demod_tile_kernel<5, 256>"

def bench_line(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith('{"metric"')][0])


bench = bench_line(os.path.join(src, "bench.json"))
json.dump(bench, open(os.path.join(out, rnd + "_bench.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "prof", "trace_kernel_stats.csv"), os.path.join(out, rnd + "_kernel_stats.csv"))

rows = [r for r in csv.DictReader(open(os.path.join(src, "prof", "trace_kernel_trace.csv"))) if KERNEL.split("<")[0] in r["Kernel_Name"]]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
prof_bench = bench_line(os.path.join(src, "prof_bench.json"))
settle, warm, steps = prof_bench["config"]["settle_steps_untimed"], prof_bench["warmup"], prof_bench["steps"]
timed = dur[settle + warm: settle + warm + steps]
summary = {
    "command": "rocprofv3 --kernel-trace --stats -S -u usec -- python3 bench.py --no-cpu   (settle %d + warmup %d + steps %d + 20 paired launches)" % (settle, warm, steps),
    "kernel": KERNEL, "launches": len(dur), "avg_ns_all_launches": sum(dur) / len(dur),
    "avg_ns_timed_region_%d_launches" % steps: sum(timed) / len(timed),
    "avg_ns_first_50_launches_clock_ramp": sum(dur[:50]) / 50, "min_ns": min(dur), "max_ns": max(dur),
    "bench_py_kernel_ms_events_region_same_run": prof_bench["roofline"]["kernel_ms_events_region"],
    "note": "trace_kernel_stats average covers every launch incl. the untimed settle phase; the timed region's average is "
            "the figure to compare with bench.py's HIP-event kernel_ms_events_region"}
json.dump(summary, open(os.path.join(out, rnd + "_kernel_trace_summary.json"), "w"), indent=1)

counters = {}
for f in glob.glob(os.path.join(pmc, "*", "*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        counters[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
waves = counters["SQ_WAVES"]["mean_per_launch"]
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
fetch_kb, write_kb = counters["FETCH_SIZE"]["mean_per_launch"], counters["WRITE_SIZE"]["mean_per_launch"]
traffic = fetch_kb * 1024 * 2 + write_kb * 1024
pm = {"command": "rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --steps 10 --warmup 2 --settle 20 --no-cpu  (scripts/gpu_pmc.sh; "
                 "one pass per counter set; never combined with --sys-trace)",
      "kernel": KERNEL, "counters": counters,
      "per_wave": {"valu": counters["SQ_INSTS_VALU"]["mean_per_launch"] / waves, "salu": counters["SQ_INSTS_SALU"]["mean_per_launch"] / waves,
                   "lds": counters["SQ_INSTS_LDS"]["mean_per_launch"] / waves, "vmem_rd": counters["SQ_INSTS_VMEM_RD"]["mean_per_launch"] / waves},
      "hbm_traffic": {"fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
                      "correction": "FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section); units of 1 KiB",
                      "bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg, "ratio_to_algorithmic": traffic / alg}}
json.dump(pm, open(os.path.join(out, rnd + "_pmc_summary.json"), "w"), indent=1)
print("bench ms/step %.4f frac %.4f | trace timed avg %.1f us (events %.1f us) | traffic x%.4f | VALU/wave %.1f SALU/wave %.1f" % (
    bench["ms_per_step"], bench["roofline"]["frac"], summary["avg_ns_timed_region_%d_launches" % steps] / 1e3,
    prof_bench["roofline"]["kernel_ms_events_region"] * 1e3, traffic / alg, pm["per_wave"]["valu"], pm["per_wave"]["salu"]))
