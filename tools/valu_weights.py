#!/usr/bin/env python3
"""Average issue cost (shader clocks on a SIMD-32 vector pipe) of the vector instructions in a kernel's hot code, from the SHIPPED code
object -- the weight scripts/summarize_bounds.py multiplies SQ_INSTS_VALU with to turn an instruction count into vector-pipe time.

Cost classes (issue cycles, tools/valubench.hip on one MI355X, profiles/archive/r04_valubench.txt; a wave64 instruction takes 2 clocks on the
SIMD-32 at full rate):
    2 clocks  v_add/sub[rev]_{u32,f32}, v_mul_f32, v_and/or/xor/not, v_lshrrev, v_ashrrev, v_mov_b32 (plain), v_bitop3_b32 -- abs / neg / clamp
              modifiers are free
    8 clocks  v_rcp_f32 (3.5 x an add)
    4 clocks  everything else on the vector pipe: v_dot*, v_fma / v_fmac, conversions, v_rndne, v_bfi, v_perm, v_alignbit, DPP moves, v_mul_lo / hi,
              v_mad_*, v_add3, v_lshl_add, v_lshlrev, min / max / med3, compares, selects, v_bfe, v_pk_* (two passes)
"Hot code" = the straight-line segments (cut at branches) that hold dot products, the f32 discriminator's rounding (v_rndne_f32) or its
integer form's reciprocal: the rounds.  The resampler pass and the prologue are a fifth of a wave's vector instructions and cheaper on
average, so the weight slightly overstates.   Usage: tools/valu_weights.py [kernel-name-regex]   -> one JSON line per kernel"""
import json, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TWO = re.compile(r"^v_(add|sub|subrev)_(u32|f32|co_u32)|^v_mul_f32|^v_(and|or|xor|not)_b32|^v_(lshrrev|ashrrev)_|^v_mov_b32_e32|^v_bitop3_b32|^v_nop")
BRANCH = re.compile(r"^(s_cbranch|s_branch|s_endpgm|s_setpc|s_swappc|s_barrier)")


def cost(ins):
    if ins.startswith("v_rcp_f32"):
        return 8
    if "_dpp" in ins or " wave_shr" in ins or " row_" in ins or " quad_perm" in ins:
        return 4
    return 2 if TWO.match(ins) else 4


def kernels(lib):
    tmp = tempfile.mkdtemp(prefix="fmd_vw_")
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True)
        out = {}
        for co in sorted(f for f in os.listdir(tmp) if "gfx950" in f):
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], cwd=tmp, check=True, capture_output=True, text=True).stdout
            sym = None
            for ln in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
                if m:
                    sym = m.group(1); out[sym] = []
                    continue
                if sym:
                    ins = ln.split("//")[0].strip()
                    if ins and not ins.startswith("<"):
                        out[sym].append(re.sub(r"\s+", " ", ins))
        names = subprocess.run(["c++filt"] + list(out), capture_output=True, text=True).stdout.splitlines()
        return {n: out[s] for s, n in zip(list(out), names)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def weight(text):
    segs, seg = [], []
    for ins in text:
        seg.append(ins)
        if BRANCH.match(ins):
            segs.append(seg); seg = []
    segs.append(seg)
    n = c = 0
    for seg in segs:
        if not any(i.startswith(("v_dot4", "v_dot2", "v_rndne_f32", "v_mfma")) for i in seg):
            continue
        for i in seg:
            if i.startswith("v_") and not i.startswith(("v_mfma", "v_readfirstlane", "v_cmpx")):
                n += 1; c += cost(i)
    return (c / n if n else 4.0), n


def short_name(demangled):
    """`void ns::kernel<args>(ParamType)` -> `ns::kernel<args>` (the form fmd_*_last_kernel / rocprofv3 rows are matched with)"""
    n = demangled[5:] if demangled.startswith("void ") else demangled
    depth = 0
    for i, ch in enumerate(n):
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0 and not n.startswith("(anonymous namespace)", i):
            return n[:i]
    return n


def weights_by_kernel():
    ks = kernels(os.path.join(ROOT, "rtl-sdr-rs_amd", "libfmd_hip.so"))
    return {short_name(n): weight(t)[0] for n, t in ks.items() if "kernel" in n}


def main():
    rx = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
    ks = kernels(os.path.join(ROOT, "rtl-sdr-rs_amd", "libfmd_hip.so"))
    for name, text in ks.items():
        if "kernel" not in name or (rx and not rx.search(name)):
            continue
        w, n = weight(text)
        print(json.dumps({"kernel": short_name(name), "valu_clocks_per_instruction": round(w, 3), "hot_vector_instructions_static": n}))


if __name__ == "__main__":
    main()
