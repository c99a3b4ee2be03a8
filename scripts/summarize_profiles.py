#!/usr/bin/env python3
"""Turn the raw outputs of scripts/gpu_check.sh <tag> and scripts/gpu_pmc.sh <tag>_pmc (under gpurun_out/) into the
committed summaries under profiles/.  Usage: summarize_profiles.py <tag> [round-prefix, default r01]"""
import collections, csv, glob, json, os, shutil, sys

def family_sha(g, tag, fam):
    """The hash of one kernel family's sources as they were on the GPU box of session `tag` (scripts/gpu_round.sh writes the file)."""
    f = os.path.join(g, tag + "_family_sha16.json")
    return json.load(open(f)).get(fam) if os.path.exists(f) else None


def summarize_firdemod(g, tag, rnd, out):
    """Fused FIR -> discriminator -> resampler kernel (config 4): bench line + PMC passes (scripts/gpu_pmc_firdemod.sh)."""
    fd = os.path.join(g, tag + "_firdemod.json")
    if os.path.exists(fd):
        shutil.copy(fd, os.path.join(out, rnd + "_config4_firdemod.json"))
    fdp = os.path.join(g, tag + "_pmc_fd", "summary.json")
    if os.path.exists(fdp):
        c = json.load(open(fdp))
        w = c["SQ_WAVES"]["mean_per_launch"]
        line = json.loads(open(fd).read().strip().splitlines()[-1]) if os.path.exists(fd) else {}
        alg = line.get("GBps", 0) * line.get("ms", 0) * 1e6
        fetch, wr = c["FETCH_SIZE"]["mean_per_launch"] * 2048, c["WRITE_SIZE"]["mean_per_launch"] * 1024
        pf = {"command": "rocprofv3 --kernel-trace --pmc <set> -- python3 tools/bench_firdemod.py (scripts/gpu_pmc_firdemod.sh; one pass per counter set)",
              "family_sha16": {"fused": family_sha(g, tag, "fused")},
              "bench_line_same_session": line, "counters": c,
              "per_wave": {k: round(c[k]["mean_per_launch"] / w, 2) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_MFMA",
                                                                            "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE") if k in c},
              "hbm_traffic": {"bytes_per_launch(2 x FETCH_SIZE + WRITE_SIZE, KiB units)": fetch + wr, "algorithmic_bytes_per_launch": round(alg),
                              "ratio": round((fetch + wr) / alg, 4) if alg else None}}
        json.dump(pf, open(os.path.join(out, rnd + "_config4_firdemod_pmc.json"), "w"), indent=1)


tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pmc, out = os.path.join(root, "gpurun_out", tag), os.path.join(root, "gpurun_out", tag + "_pmc"), os.path.join(root, "profiles")
if "--firdemod-only" in sys.argv:      # a session that re-measured only the fused FIR kernel (its source is not one of the headline's)
    summarize_firdemod(os.path.join(root, "gpurun_out"), tag, rnd, out)
    sys.exit(0)
KERNEL = "fmd_demod_tile_kernel<5, 2>"
sys.path.insert(0, root)
import bench as _bench   # kernel_source_hash(): ties the PMC summary to the sources it was measured on

def bench_line(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith('{"metric"')][0])


bench = bench_line(os.path.join(src, "bench.json"))
json.dump(bench, open(os.path.join(out, rnd + "_bench.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "prof", "trace_kernel_stats.csv"), os.path.join(out, rnd + "_kernel_stats.csv"))

rows = [r for r in csv.DictReader(open(os.path.join(src, "prof", "trace_kernel_trace.csv"))) if KERNEL.split("<")[0] in r["Kernel_Name"]]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
prof_bench = bench_line(os.path.join(src, "prof_bench.json"))
settle, warm, steps = prof_bench["config"]["settle_steps_untimed"], prof_bench["warmup"], prof_bench["steps"]
regions = prof_bench.get("timing", {}).get("regions", 1)
timed = dur[settle + warm: settle + warm + steps * regions]
summary = {
    "command": "rocprofv3 --kernel-trace --stats -S -u usec -- python3 bench.py --no-cpu --no-extra   (settle %d + warmup %d + %d regions x %d steps + 40 paired launches)" % (settle, warm, regions, steps),
    "kernel": KERNEL, "launches": len(dur), "avg_ns_all_launches": sum(dur) / len(dur),
    "avg_ns_timed_regions_%d_launches" % len(timed): sum(timed) / len(timed),
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
# the hash written on the GPU box next to the counters (scripts/gpu_pmc.sh); a session without it cannot be tied to sources
hash_file = os.path.join(pmc, "kernel_source_sha16.txt")
measured_hash = open(hash_file).read().strip() if os.path.exists(hash_file) else None
if measured_hash != _bench.kernel_source_hash():
    print("WARNING: the kernel sources have changed since the PMC passes (%s measured, %s now): bench.py will not quote this traffic" % (
        measured_hash, _bench.kernel_source_hash()))
waves = counters["SQ_WAVES"]["mean_per_launch"]
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
fetch_kb, write_kb = counters["FETCH_SIZE"]["mean_per_launch"], counters["WRITE_SIZE"]["mean_per_launch"]
traffic = fetch_kb * 1024 * 2 + write_kb * 1024
pm = {"command": "rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --steps 10 --warmup 2 --settle 20 --no-cpu --no-extra  (scripts/gpu_pmc.sh; "
                 "one pass per counter set; never combined with --sys-trace)",
      "kernel": KERNEL, "kernel_source_sha16": measured_hash, "family_sha16": {"tile_even": measured_hash}, "counters": counters,
      "per_wave": {"valu": counters["SQ_INSTS_VALU"]["mean_per_launch"] / waves, "salu": counters["SQ_INSTS_SALU"]["mean_per_launch"] / waves,
                   "lds": counters["SQ_INSTS_LDS"]["mean_per_launch"] / waves, "vmem_rd": counters["SQ_INSTS_VMEM_RD"]["mean_per_launch"] / waves},
      "hbm_traffic": {"fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
                      "correction": "FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section); units of 1 KiB",
                      "bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg, "ratio_to_algorithmic": traffic / alg}}
json.dump(pm, open(os.path.join(out, rnd + "_pmc_summary.json"), "w"), indent=1)
print("bench ms/step %.4f frac %.4f | trace timed avg %.1f us (events %.1f us) | traffic x%.4f | VALU/wave %.1f SALU/wave %.1f" % (
    bench["ms_per_step"], bench["roofline"]["frac"], summary["avg_ns_timed_regions_%d_launches" % len(timed)] / 1e3,
    prof_bench["roofline"]["kernel_ms_events_region"] * 1e3, traffic / alg, pm["per_wave"]["valu"], pm["per_wave"]["salu"]))

# ---- config 4 (FIR): <tag>_fir_mfma.json, <tag>_fir_valu.json, <tag>_fir/prof, <tag>_pmc_fir/summary.json ----------
g = os.path.join(root, "gpurun_out")
if os.path.exists(os.path.join(g, tag + "_fir_mfma.json")):
    m = json.loads(open(os.path.join(g, tag + "_fir_mfma.json")).read().strip().splitlines()[-1])
    v = json.loads(open(os.path.join(g, tag + "_fir_valu.json")).read().strip().splitlines()[-1])
    shutil.copy(os.path.join(g, tag + "_fir", "prof", "trace_kernel_stats.csv"), os.path.join(out, rnd + "_config4_fir_kernel_stats.csv"))
    rows = [r for r in csv.DictReader(open(os.path.join(g, tag + "_fir", "prof", "trace_kernel_trace.csv"))) if "fir_mfma" in r["Kernel_Name"]]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    fir = {"matrix_core_form": m, "valu_form_FMD_FIR_MFMA_0": v,
           "rocprofv3_kernel_trace": {"kernel": rows[0]["Kernel_Name"], "launches": len(d), "avg_us_all": round(sum(d) / len(d), 2),
                                      "avg_us_timed_region_last_100": round(sum(d[-100:]) / 100, 2)},
           "ablations_exp_build_ms_before_nt_loads": {"full": 0.145, "no_mfma_phase": 0.139, "no_loads": 0.081, "no_stores": 0.100, "stores_only": 0.045},
           "history": "VALU form 0.307 ms -> MFMA form with register staging 0.173 -> LDS-DMA staging 0.146 -> history write folded "
                      "into the last tile 0.144 -> nt policy on the staging loads 0.137 (same-box A/B: -5 %)"}
    json.dump(fir, open(os.path.join(out, rnd + "_config4_fir.json"), "w"), indent=1)
    c = json.load(open(os.path.join(g, tag + "_pmc_fir", "summary.json")))
    alg = m["algorithmic_GBps"] * m["ms_per_call"] * 1e6
    fetch, wr = c["FETCH_SIZE"]["mean_per_launch"] * 2048, c["WRITE_SIZE"]["mean_per_launch"] * 1024
    w = c["SQ_WAVES"]["mean_per_launch"]
    pf = {"command": "rocprofv3 --kernel-trace --pmc <set> -- python3 tools/bench_fir.py (scripts/gpu_pmc_fir.sh; one pass per counter set)",
          "kernel": rows[0]["Kernel_Name"], "counters": c,
          "per_wave": {k: round(c[k]["mean_per_launch"] / w, 2) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_MFMA")},
          "mfma_busy_cycles_per_mfma": round(c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / c["SQ_INSTS_MFMA"]["mean_per_launch"], 2),
          "lds_bank_conflict_cycles": c["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"],
          "lds_bank_conflict_cycles_with_FMD_FIR_SWZ_1": 0.0,   # measured; that layout is off by default (1.4 % slower)
          "hbm_traffic": {"fetch_bytes_per_launch(FETCH_SIZE x 1 KiB x 2)": fetch, "write_bytes_per_launch(WRITE_SIZE x 1 KiB)": wr,
                          "bytes_per_launch": fetch + wr, "algorithmic_bytes_per_launch": round(alg), "ratio": round((fetch + wr) / alg, 4)}}
    json.dump(pf, open(os.path.join(out, rnd + "_config4_fir_pmc.json"), "w"), indent=1)
    print("FIR mfma %.4f ms (%.1f %% of HBM spec), valu %.4f ms | trace timed %.1f us | traffic x%.4f | bank conflicts %.0f" % (
        m["ms_per_call"], 100 * m["hbm_frac_of_8TBps"], v["ms_per_call"], fir["rocprofv3_kernel_trace"]["avg_us_timed_region_last_100"],
        pf["hbm_traffic"]["ratio"], pf["lds_bank_conflict_cycles"]))
# ---- config 4, stand-alone FIR, round 4 onwards: <tag>_fir.json (tools/bench_fir.py) + <tag>_pmc_fir/summary.json -------------------
fj, fs = os.path.join(g, tag + "_fir.json"), os.path.join(g, tag + "_pmc_fir", "summary.json")
if os.path.exists(fj) and os.path.exists(fs) and not os.path.exists(os.path.join(g, tag + "_fir_mfma.json")):
    m = json.loads(open(fj).read().strip().splitlines()[-1])
    c = json.load(open(fs))
    w = c["SQ_WAVES"]["mean_per_launch"]
    alg = m["algorithmic_GBps"] * m["ms_per_call"] * 1e6
    fetch, wr = c["FETCH_SIZE"]["mean_per_launch"] * 2048, c["WRITE_SIZE"]["mean_per_launch"] * 1024
    pf = {"command": "rocprofv3 --kernel-trace --pmc <set> -- python3 tools/bench_fir.py (scripts/gpu_pmc_fir.sh; one pass per counter set)",
          "kernel": c.get("kernel_name"), "kernel_ns_under_pmc": c.get("kernel_ns_under_pmc"), "bench_line_same_session": m,
          "family_sha16": {"fir": family_sha(g, tag, "fir")},
          "counters": {k: v for k, v in c.items() if isinstance(v, dict) and "mean_per_launch" in v},
          "per_wave": {k: round(c[k]["mean_per_launch"] / w, 2) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_MFMA",
                                                                        "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if k in c},
          "mfma_busy_cycles_per_mfma": round(c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / c["SQ_INSTS_MFMA"]["mean_per_launch"], 2),
          "lds_bank_conflict_cycles": c["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"], "lds_idx_active_cycles": c["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"],
          "hbm_traffic": {"fetch_bytes_per_launch(FETCH_SIZE x 1 KiB x 2)": fetch, "write_bytes_per_launch(WRITE_SIZE x 1 KiB)": wr,
                          "bytes_per_launch": fetch + wr, "algorithmic_bytes_per_launch": round(alg), "ratio": round((fetch + wr) / alg, 4)}}
    json.dump(pf, open(os.path.join(out, rnd + "_config4_fir_pmc.json"), "w"), indent=1)
    print("FIR %.4f ms (%.1f %% of HBM spec) | traffic x%.4f | bank conflicts %.0f of %.0f LDS cycles" % (
        m["ms_per_call"], 100 * m["hbm_frac_of_8TBps"], pf["hbm_traffic"]["ratio"], pf["lds_bank_conflict_cycles"], pf["lds_idx_active_cycles"]))
cfgs = os.path.join(g, tag + "_configs.jsonl")
if os.path.exists(cfgs):
    with open(os.path.join(out, rnd + "_configs.jsonl"), "w") as f:
        f.writelines(l for l in open(cfgs) if l.startswith('{"config"'))
# per-configuration PMC (scripts/gpu_pmc_configs.sh): three passes per configuration
pc = os.path.join(g, tag + "_pmc_configs.jsonl")
if os.path.exists(pc):
    shutil.copy(pc, os.path.join(out, rnd + "_pmc_configs.jsonl"))
pr = os.path.join(g, tag + "_pmc_regions.jsonl")          # per-region instruction counts (scripts/gpu_pmc_regions.sh)
if os.path.exists(pr):
    shutil.copy(pr, os.path.join(out, rnd + "_pmc_regions.jsonl"))
pfr = os.path.join(g, tag + "_pmc_fd_regions.jsonl")       # per-region counters of the fused FIR kernel (scripts/gpu_pmc_fd_regions.sh)
if os.path.exists(pfr):
    shutil.copy(pfr, os.path.join(out, rnd + "_pmc_fd_regions.jsonl"))
summarize_firdemod(g, tag, rnd, out)
