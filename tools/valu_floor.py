#!/usr/bin/env python3
"""Vector instructions per discriminator sample in the round loops of the SHIPPED kernels, by what they are for (an
instruction floor for the vector-bound rows).  From the code object inside libfmd_hip.so: the first full-round loop of each kernel
(two discriminator samples per lane and trip: two v_rcp_f32) is cut into
    window     u8 -> s8 flips and byte dot products of the two windows (v_xor 0x80808080, v_dot4*, v_alignbit)
    to_f32     biased sum -> f32 (v_add_f32 0xcb400000: one per component)
    moves      predecessor from the neighbouring lane (DPP) and packs
    product    a * conj(b): four fmas per sample
    atan2      fast_atan2 in f32 incl. the store conversion (19 per sample)
    other      address arithmetic, predication -- what is neither the window sum, the product nor fast_atan2
Usage: tools/valu_floor.py [kernel-regex]   -> one JSON line per kernel (per round body and per sample)"""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import valu_weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = r"fmd_demod_tile_kernel<(-1|-3|-5|3|5), 2>|fmd_demod_stream_kernel<[12], 2>"


def classify(ins):
    op = ins.split()[0]
    if not op.startswith("v_") or op.startswith(("v_readfirstlane", "v_cmpx")):
        return None
    if "dpp" in ins or " wave_shr" in ins or " wave_ror" in ins or op.startswith("v_perm"):
        return "moves"
    if op.startswith("v_dot4") or op.startswith("v_alignbit") or (op.startswith("v_xor_b32") and "0x80808080" in ins):
        return "window"
    if op.startswith("v_add_f32") and "0xcb400000" in ins:
        return "to_f32"
    if op.startswith(("v_fma_f32", "v_fmac_f32")) and "clamp" not in ins:
        return "product"
    if op.startswith(("v_sub_f32", "v_subrev_f32", "v_add_f32", "v_mul_f32", "v_rcp_f32", "v_rndne_f32", "v_fma_f32", "v_bfi_b32", "v_bitop3_b32", "v_cvt_i32_f32")):
        return "atan2"
    return "other"


LOOP_EDGE = ("s_cbranch_scc", "s_branch", "s_barrier", "s_endpgm", "s_cbranch_vcc")


def round_body(text):
    """instructions of the first full round of a wave: the stretch between two loop edges (s_cbranch_scc* / s_branch: the round
    loops' trip control runs on the scalar unit; s_cbranch_execz inside is the predicated store of a lane's first sample) that holds
    exactly two v_rcp_f32 and the windows' dot products"""
    edges = [-1] + [i for i, ins in enumerate(text) if ins.startswith(LOOP_EDGE)] + [len(text)]
    for a, b in zip(edges, edges[1:]):
        body = text[a + 1:b + 1]
        if sum(i.startswith("v_rcp_f32") for i in body) == 2 and sum(i.startswith("v_dot4") for i in body) >= 4:
            return body
    return []


def main():
    rx = re.compile(sys.argv[1] if len(sys.argv) > 1 else DEFAULT)
    ks = valu_weights.kernels(os.path.join(ROOT, "rtl-sdr-rs_amd", "libfmd_hip.so"))
    for name, text in ks.items():
        short = valu_weights.short_name(name)
        if not rx.search(short):
            continue
        body = round_body(text)
        mix = {}
        for ins in body:
            c = classify(ins)
            if c:
                mix[c] = mix.get(c, 0) + 1
        tot = sum(mix.values())
        core = mix.get("window", 0) + mix.get("to_f32", 0) + mix.get("product", 0) + mix.get("atan2", 0)
        print(json.dumps({"kernel": short, "round_body_valu": tot, "by_class": mix,
                          "per_sample": {k: round(v / 2.0, 1) for k, v in mix.items()},
                          "window_product_atan2_share": round(core / tot, 3) if tot else None,
                          "overhead_share_in_the_rounds": round(1 - core / tot, 3) if tot else None}))


if __name__ == "__main__":
    main()
