#!/usr/bin/env python3
"""profiles/<round>_bounds.json: for every measured row (demodulation configurations, the stand-alone FIR, the fused FIR kernel) how
busy the vector / scalar / matrix pipes and the LDS were, from the committed PMC passes -- the figures bench.py attaches to its
`extra` rows (`bound`, `valu_issue_frac`, ...).  Usage: summarize_bounds.py <tag> <round>     (reads gpurun_out/<tag>_pmc_configs.jsonl,
gpurun_out/<tag>_pmc_fir/summary.json, gpurun_out/<tag>_pmc_fd/summary.json; method in profiles/README.md)

Units (MI355X, gfx950): SQ_BUSY_CYCLES counts shader clocks while the launch runs, summed over the chip's 32 shader engines; SQ_INSTS_* are
wave-instructions.  SQ_ACTIVE_INST_VALU / _SCA turned out to be issue-SLOT counts (1.02 / 1.08 quad-cycles per instruction whatever the
instruction: session r05g), not pipe time -- a SIMD-32 starts a 2-clock instruction every 2 clocks, so slots x 4 clocks exceeds the launch
(1.39 at downsample 1).  Pipe time is therefore instruction count x issue cost.  With 1024 SIMDs (256 CUs x 4):
    launch_cycles   = SQ_BUSY_CYCLES / 32
    valu_issue_frac = SQ_INSTS_VALU x w / 1024 / launch_cycles    w = average issue clocks per vector instruction of the kernel's rounds, from the
                      SHIPPED code object (tools/valu_weights.py: 2 clocks add / logic / f32 add-mul, 8 v_rcp_f32, 4 everything else; ~3.0 here)
    salu_issue_frac = SQ_INSTS_SALU x 4.7 / 1024 / launch_cycles  (an s_add_u32 costs a SIMD 2.23 ns = 4.7 clocks at 8 waves: profiles/archive/r04_salubench.txt)
    mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / launch_cycles   (shader clocks the matrix pipes were busy, summed over SIMDs)
`bound` in bench.py = the largest of (HBM fraction of the launch, valu_issue_frac, salu_issue_frac, mfma_busy_frac)."""
import json, os, sys
tag, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = os.path.join(root, "gpurun_out")
N_SE, N_SIMD, SALU_CLOCKS = 32.0, 1024.0, 4.7
sys.path.insert(0, os.path.join(root, "tools"))
import valu_weights
W = valu_weights.weights_by_kernel()


def kernel_weight(name):
    return W.get((name or "").strip(), 3.0)
rows, sha = {}, None
for l in open(os.path.join(g, tag + "_pmc_configs.jsonl")):
    d = json.loads(l)
    if "per_wave" not in d or not d.get("cfg"):
        continue
    key = ",".join(str(x) for x in d["cfg"])
    r = rows.setdefault(key, {"config": d["config"], "kernel": d.get("kernel_reported"), "waves_per_launch": d["waves_per_launch"], "per_wave": {}})
    r["per_wave"].update(d["per_wave"])
    sha = d.get("kernel_source_sha16", sha)
# the hashes of every kernel family's sources AS MEASURED (written on the GPU box by scripts/gpu_round.sh): bench.py quotes a row only
# while its family still hashes the same
fam_file = os.path.join(g, tag + "_family_sha16.json")
fam = json.load(open(fam_file)) if os.path.exists(fam_file) else {}
out = {"kernel_source_sha16": sha, "family_sha16": fam, "method": __doc__.split("Units", 1)[1].strip(), "demod": {}}
for key, r in rows.items():
    p, w = r["per_wave"], r["waves_per_launch"]
    if "SQ_BUSY_CYCLES" not in p or "SQ_ACTIVE_INST_VALU" not in p or "SQ_INSTS_VALU" not in p:
        continue
    cyc = p["SQ_BUSY_CYCLES"] * w / N_SE
    kw = kernel_weight(r["kernel"])
    o = {"config": r["config"], "kernel": r["kernel"], "launch_cycles_under_pmc": round(cyc), "valu_clocks_per_instruction": round(kw, 3),
         "valu_issue_frac": round(p["SQ_INSTS_VALU"] * w * kw / N_SIMD / cyc, 3),
         "salu_issue_frac": round(p.get("SQ_INSTS_SALU", 0) * w * SALU_CLOCKS / N_SIMD / cyc, 3),
         "valu_issue_slots_x4_frac": round(p["SQ_ACTIVE_INST_VALU"] * w * 4 / N_SIMD / cyc, 3),
         "valu_per_wave": p.get("SQ_INSTS_VALU"), "salu_per_wave": p.get("SQ_INSTS_SALU"),
         "lds_bank_conflict_share": round(p["SQ_LDS_BANK_CONFLICT"] / p["SQ_LDS_IDX_ACTIVE"], 3) if p.get("SQ_LDS_IDX_ACTIVE") else None}
    out["demod"][key] = o
# ---- the compute side ALONE (scripts/gpu_pmc_alone.sh): shader clocks of the experiment build with the staging loads ablated over the
# shader clocks of the same build unablated -- clock-independent, needs no price list, never above 1 by construction
alone_file = os.path.join(g, tag + "_pmc_alone.jsonl")
if os.path.exists(alone_file):
    cyc = {}
    for l in open(alone_file):
        d = json.loads(l)
        if "SQ_BUSY_CYCLES" in d.get("per_launch", {}):
            cyc.setdefault(d["kernel"], {})[d["dbg"]] = d["per_launch"]["SQ_BUSY_CYCLES"]
    for o in out["demod"].values():
        c = cyc.get(o["kernel"])
        if c and 0 in c and 16 in c and "stream_kernel" not in o["kernel"]:      # (the streaming kernel stages nothing: the ablation does not apply)
            o["compute_alone_cycles_frac"] = round(c[16] / c[0], 3)
# ---- calibration of the price list: the instruction-count model prices a vector instruction at the issue clocks of
# tools/valu_weights.py's classes; a pipe cannot be more than 100 % busy, so the row that comes out highest caps the price -- every
# row's vector fraction is scaled by the same factor so that it reads 1.00 (the weights overstate by that much: 2 % in round 5)
top = max((o["valu_issue_frac"] for o in out["demod"].values()), default=0.0)
scale = 1.0 / top if top > 1.0 else 1.0
out["valu_weight_scale"] = round(scale, 4)
if scale != 1.0:
    for o in out["demod"].values():
        o["valu_issue_frac_uncalibrated"] = o["valu_issue_frac"]
        o["valu_issue_frac"] = round(o["valu_issue_frac"] * scale, 3)
for name, sub in (("config4_fir", tag + "_pmc_fir"), ("config4_fir_demod_fused", tag + "_pmc_fd")):
    f = os.path.join(g, sub, "summary.json")
    if not os.path.exists(f):
        continue
    c = {k: v["mean_per_launch"] for k, v in json.load(open(f)).items() if isinstance(v, dict) and "mean_per_launch" in v}
    if "SQ_BUSY_CYCLES" not in c:
        continue
    cyc = c["SQ_BUSY_CYCLES"] / N_SE
    o = {"launch_cycles_under_pmc": round(cyc), "waves_per_launch": round(c.get("SQ_WAVES", 0))}
    kname = {"config4_fir": "(anonymous namespace)::fmd_fir_mfma_kernel<6, false, 3>", "config4_fir_demod_fused": "(anonymous namespace)::fmd_firdemod_regs_kernel<6, 8, true>"}[name]
    kw = kernel_weight(kname) * scale
    o["kernel"], o["valu_clocks_per_instruction"] = kname, round(kw, 3)
    if "SQ_INSTS_VALU" in c:
        o["valu_issue_frac"] = round(c["SQ_INSTS_VALU"] * kw / N_SIMD / cyc, 3)
    if "SQ_INSTS_SALU" in c:
        o["salu_issue_frac"] = round(c["SQ_INSTS_SALU"] * SALU_CLOCKS / N_SIMD / cyc, 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        o["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / N_SIMD / cyc, 3)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        o["lds_bank_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 3)
    w = c.get("SQ_WAVES")
    if w:
        o["per_wave"] = {k: round(c[k] / w, 1) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA") if k in c}
    out[name] = o
json.dump(out, open(os.path.join(root, "profiles", rnd + "_bounds.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
