#!/usr/bin/env python3
"""Per-kernel digest of the gfx950 ISA hipcc emits for the library's translation units (no GPU needed).

    tools/isa_digest.py [--src fmd_tile_lds_even.hip ...] [--kernel REGEX] [--blocks] [--json out.json]

For every kernel (and out-of-line device function) of the given sources: a hash of its instruction stream with
symbol names and local labels normalised -- two builds whose digests agree run the same code, which is how a refactoring
of the sources is checked to be codegen-neutral -- the static instruction mix (VALU / SALU / LDS / VMEM / SMEM / MFMA /
branch), the resources the kernel descriptor asks for (VGPRs, SGPRs, scratch bytes, occupancy) and, with --blocks, the
same mix per basic block with its loop depth (the static side of the per-region breakdown in profiles/)."""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rtl-sdr-rs_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"]
DEFAULT = ["fmd_tile_lds_even.hip", "fmd_tile_lds_wide.hip", "fmd_tile_lds_odd.hip", "fmd_tile_stream.hip",
           "fmd_generic_kernel.hip", "fmd_fir.hip", "fmd_firdemod.hip"]


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_store"):
        return "smem"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op in ("s_endpgm", "s_swappc_b64", "s_setpc_b64"):
        return "branch"
    if op in ("s_waitcnt", "s_nop", "s_barrier", "s_setprio", "s_sleep"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    return "other"


def emit_asm(src, extra):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-x", "hip", "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out]
    subprocess.run(cmd, check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def demangle(names):
    if not names:
        return {}
    p = subprocess.run(["c++filt"] + names, capture_output=True, text=True)
    return dict(zip(names, p.stdout.splitlines()))


def functions(asm):
    """yield (symbol, body lines, descriptor dict)"""
    lines = asm.splitlines()
    i, n = 0, len(lines)
    while i < n:
        m = re.match(r"^([_A-Za-z][\w$.]*):\s*(;.*)?$", lines[i])
        if m and not lines[i].startswith(".L"):
            sym, body = m.group(1), []
            i += 1
            while i < n and not lines[i].startswith(".Lfunc_end"):
                body.append(lines[i])
                i += 1
            desc = {}
            j = i
            while j < n and j < i + 400 and not re.match(r"^[_A-Za-z][\w$.]*:\s*(;.*)?$", lines[j]):
                for key, pat in (("vgpr", r"; NumVgprs: (\d+)"), ("agpr", r"; NumAgprs: (\d+)"), ("sgpr", r"; NumSgprs: (\d+)"),
                                 ("scratch", r"; ScratchSize: (\d+)"), ("occupancy", r"; Occupancy: (\d+)"), ("lds", r"; LDSByteSize: (\d+)")):
                    mm = re.search(pat, lines[j])
                    if mm and key not in desc:
                        desc[key] = int(mm.group(1))
                j += 1
            yield sym, body, desc
        else:
            i += 1


def digest(body, sym):
    """normalised instruction stream: comments dropped, local labels renumbered in order of appearance, own symbol masked"""
    labels, norm, blocks = {}, [], []
    cur = {"label": "entry", "depth": 0, "mix": {}}
    for ln in body:
        depth = None
        if ";" in ln:
            mm = re.search(r"Depth=(\d+)", ln)
            if mm:
                depth = int(mm.group(1))
            ln = ln.split(";", 1)[0]
        ln = ln.strip()
        if not ln or ln.startswith(".") and not ln.startswith(".LBB"):
            continue
        lm = re.match(r"^(\.LBB\d+_\d+):$", ln)
        if lm:
            labels.setdefault(lm.group(1), "L%d" % len(labels))
            norm.append(labels[lm.group(1)] + ":")
            cur = {"label": labels[lm.group(1)], "depth": depth or 0, "mix": {}}
            blocks.append(cur)
            continue
        ln = re.sub(r"\.LBB\d+_\d+", lambda m_: labels.setdefault(m_.group(0), "L%d" % len(labels)), ln)
        ln = ln.replace(sym, "SELF")
        ln = re.sub(r"_ZN[\w$.]+", "SYM", ln)            # callees (polar_f64, exc_emit): names differ per namespace
        norm.append(re.sub(r"\s+", " ", ln))
        op = ln.split()[0]
        k = classify(op)
        cur["mix"][k] = cur["mix"].get(k, 0) + 1
        if not blocks:
            blocks.append(cur)
    mix = {}
    for b in blocks:
        for k, v in b["mix"].items():
            mix[k] = mix.get(k, 0) + v
    return hashlib.sha256("\n".join(norm).encode()).hexdigest()[:16], mix, blocks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", nargs="*", default=DEFAULT)
    ap.add_argument("--asm", nargs="*", default=[], help="already emitted .s files instead of compiling --src")
    ap.add_argument("--kernel", default="", help="only symbols whose demangled name matches this regex")
    ap.add_argument("--blocks", action="store_true", help="per basic block mix (label, loop depth)")
    ap.add_argument("--define", "-D", action="append", default=[])
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    texts = [(f, open(f).read()) for f in a.asm] if a.asm else [(s, emit_asm(s, ["-D" + d for d in a.define])) for s in a.src]
    res = []
    for src, asm in texts:
        fs = list(functions(asm))
        names = demangle([f[0] for f in fs])
        for sym, body, desc in fs:
            name = names.get(sym, sym)
            if a.kernel and not re.search(a.kernel, name):
                continue
            h, mix, blocks = digest(body, sym)
            rec = {"src": os.path.basename(src), "name": name, "sha16": h, "static_mix": mix, **desc}
            if a.blocks:
                rec["blocks"] = [b for b in blocks if b["mix"]]
            res.append(rec)
    for r in res:
        line = {k: v for k, v in r.items() if k != "blocks"}
        print(json.dumps(line))
        for b in r.get("blocks", []):
            print("    %-6s depth %d  %s" % (b["label"], b["depth"], " ".join("%s=%d" % kv for kv in sorted(b["mix"].items()))))
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
