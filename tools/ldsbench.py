#!/usr/bin/env python3
"""What the LDS read patterns of the matrix-core kernels cost on gfx950 (tools/ldsbench.hip replays them): the operand reads of the
sparse forms (two 16-byte reads per lane and 128-byte K chunk) as shipped in round 5, and the candidates that keep every group of
lanes the LDS serves together on distinct 16-byte slots of the 256-byte bank row.  Lane = j + 16 q (column j, K quarter q).
Usage: tools/ldsbench.py [--json out.jsonl]"""
import argparse, ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "libldsbench.so")


def lanes():
    for l in range(64):
        yield l, l & 15, l >> 4


def patterns():
    P = {}
    P["contiguous (lane l at 16 l): the conflict-free yardstick"] = (16, lambda l, j, q, h: 16 * l + 1024 * h)
    # fused FIR kernel, sparse forms (fmd_firdemod.hip fd_reg_body<SP = true>): columns of PC = 4 NG - 2 outputs of 16 bytes
    for ng in (4, 6, 8):
        pc = 16 * (4 * ng - 2)
        P["fused sparse NG=%d, round 5 (col %d j + 32 q + 16 h)" % (ng, pc)] = (16, lambda l, j, q, h, pc=pc: pc * j + 32 * q + 16 * h)
        P["fused sparse NG=%d, halves swapped in odd K quarters" % ng] = (16, lambda l, j, q, h, pc=pc: pc * j + 32 * q + 16 * (h ^ (q & 1)))
        P["fused sparse NG=%d, halves swapped in K quarters 1, 2" % ng] = (16, lambda l, j, q, h, pc=pc: pc * j + 32 * q + 16 * (h ^ ((q ^ (q >> 1)) & 1)))
        P["fused sparse NG=%d, halves swapped in K quarters 2, 3" % ng] = (16, lambda l, j, q, h, pc=pc: pc * j + 32 * q + 16 * (h ^ (q >> 1)))
    # fused dense (round 4) for reference: one 16-byte read per lane and 64-byte chunk
    P["fused dense NG=8 (col 480 j + 16 q; 64 h = next fragment)"] = (16, lambda l, j, q, h: 480 * j + 16 * q + 64 * h)
    # stand-alone FIR, sparse form (fmd_fir.hip DIGITS = 3): columns of 8 outputs = 128 bytes
    P["FIR sparse, round 5 (128 j + 32 q + 16 h)"] = (16, lambda l, j, q, h: 128 * j + 32 * q + 16 * h)
    P["FIR sparse, halves swapped in odd K quarters"] = (16, lambda l, j, q, h: 128 * j + 32 * q + 16 * (h ^ (q & 1)))
    # ... with the tile image skewed by DMA lane permutation: 16-byte slot s of column j stored at slot s ^ f(j)
    for name, f in (("s ^ (j >> 1)", lambda j: (j >> 1) & 7), ("s ^ (j & 7)", lambda j: j & 7), ("s ^ 2 (j & 3)", lambda j: 2 * (j & 3)),
                    ("s ^ (j >> 1 & 3) * 2", lambda j: 2 * ((j >> 1) & 3)), ("s ^ ((j >> 1) & 1) * 4 ^ ((j >> 2) & 1) * 2", lambda j: 4 * ((j >> 1) & 1) ^ 2 * ((j >> 2) & 1))):
        P["FIR sparse, slot %s" % name] = (16, lambda l, j, q, h, f=f: 128 * j + 16 * (((2 * q + h) ^ f(j)) & 7))
        P["FIR sparse, slot %s, halves swapped in odd K quarters" % name] = (16, lambda l, j, q, h, f=f: 128 * j + 16 * (((2 * q + (h ^ (q & 1))) ^ f(j)) & 7))
    # stand-alone FIR dense two-digit form (col 64 j + 16 q), as shipped and with round 2's swizzle
    P["FIR dense (64 j + 16 q; 64 h = next chunk)"] = (16, lambda l, j, q, h: 64 * j + 16 * q + 64 * h)
    # 8-byte reads: four per lane and chunk
    P["contiguous 8-byte reads (lane l at 8 l): the yardstick of the 8-byte rows"] = (8, lambda l, j, q, h: 8 * l + 512 * h)
    P["FIR sparse as 8-byte reads (128 j + 32 q + 8 h)"] = (8, lambda l, j, q, h: 128 * j + 32 * q + 8 * h)
    # service-order probes: which lanes does the LDS serve together?  two lanes on one slot, everyone else spread
    return P


def probe_pairs():
    """(a, b): lanes a and b read the SAME 16-byte slot, every other lane its own -- if a and b are served in different passes the
    read is conflict-free.  Gives the pass membership of ds_read_b128 by experiment."""
    out = {}
    for b in (1, 2, 3, 4, 8, 12, 15, 16, 20, 28, 31, 32, 48, 63):
        out["probe: lane %d shares lane 0's slot" % b] = (16, lambda l, j, q, h, b=b: 16 * (0 if l == b else l) + 1024 * h)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--iters", type=int, default=4000)
    a = ap.parse_args()
    if not os.path.exists(SO):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(ROOT, "tools", "ldsbench.hip"), "-o", SO])
    lib = C.CDLL(SO)
    lib.ldsbench_run.restype = C.c_float
    lib.ldsbench_run.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int]
    blocks, reps = 256 * 5, 5
    rows, base = [], None
    allp = list(patterns().items()) + list(probe_pairs().items())
    for name, (width, f) in allp:
        a0 = (C.c_int * 64)(*[f(l, j, q, 0) for l, j, q in lanes()])
        a1 = (C.c_int * 64)(*[f(l, j, q, 1) for l, j, q in lanes()])
        best = min(lib.ldsbench_run(a0, a1, width, a.iters, blocks, reps) for _ in range(3))
        assert best > 0, (name, best)
        # read instructions per CU: blocks / 256 CUs x 4 waves x iters x 4 reads x reps
        n = blocks / 256.0 * 4 * a.iters * 4 * reps
        ns = best * 1e6 / n
        base = base or ns
        # distinct 16-byte slots (mod 256 bytes) touched per 16 consecutive lanes -- a static hint only; the measured ratio is what counts
        row = {"pattern": name, "width": width, "ns_per_read_per_cu": round(ns, 3), "vs_contiguous": round(ns / base, 3)}
        rows.append(row)
        print(json.dumps(row), flush=True)
    if a.json:
        with open(a.json, "w") as fh:
            for r in rows:
                fh.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
