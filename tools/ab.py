#!/usr/bin/env python3
"""Interleaved A/B of library variants inside ONE process on ONE box (same clocks, same memory): every variant is a
set of environment variables read when a handle is created (FMD_FAST, FMD_KT, FMD_XCD, ...) or a different
library build (LIB=path is handled by running this tool once per build).  Usage:
    tools/ab.py [--cfg ref|24|D,fast,slow]... [--rounds 3] [--steps 100] name:VAR=val,VAR=val name2: ...
Prints one line per (config, variant) with the per-round times and their median."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtl_sdr_rs_amd as fmd

NAMED = {"ref": (6, 170000, 32000), "24": (10, 240000, 32000)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", action="append", default=[])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--settle", type=int, default=150)
    ap.add_argument("--channels", type=int, default=4096)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    cfgs = [NAMED[c] if c in NAMED else tuple(int(x) for x in c.split(",")) for c in (a.cfg or ["24"])]
    variants = []
    for v in a.variants:
        name, _, envs = v.partition(":")
        variants.append((name, dict(e.split("=", 1) for e in envs.split(",") if e)))
    nch, N = a.channels, fmd.DEFAULT_BUF_LENGTH
    stream = torch.cuda.current_stream().cuda_stream
    bufs = []
    for b in range(3):
        t = torch.empty((nch, N), dtype=torch.uint8, device="cuda")
        fmd.synth.fill_device(t.data_ptr(), nch, N, sample_offset=b * (N // 2), stream=stream)
        bufs.append(t)
    knobs = sorted({k for _, e in variants for k in e})
    for D, fast, slow in cfgs:
        cfg = fmd.DemodConfig(fast, fast, slow, D, max(1, (1 << 15) // (128 * D)))
        res = {name: [] for name, _ in variants}
        alg = None
        for rnd in range(a.rounds):
            for name, env in variants:
                for k in knobs:
                    os.environ.pop(k, None)
                os.environ.update(env)
                bank = fmd.DemodBank(cfg, nch)
                cap = bank.out_cap(N)
                out = torch.zeros((nch, cap), dtype=torch.int16, device="cuda")
                for i in range(a.settle):
                    bank.demodulate_device(bufs[i % 3].data_ptr(), N, out.data_ptr(), cap, None, stream)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(a.steps):
                    bank.demodulate_device(bufs[i % 3].data_ptr(), N, out.data_ptr(), cap, None, stream)
                e1.record(); torch.cuda.synchronize()
                bank.check()
                res[name].append(e0.elapsed_time(e1) / a.steps)
                alg = nch * N + 2 * int(bank.last_out_len().sum())
                tiling = bank.tiling()
                bank.close(); del out
        for name, _ in variants:
            ts = sorted(res[name]); med = ts[len(ts) // 2]
            print(json.dumps({"cfg": [D, fast, slow], "variant": name, "ms": [round(t, 4) for t in res[name]], "median_ms": round(med, 4),
                              "frac": round(alg / med / 1e6 / 8000, 4), "tiling": tiling}), flush=True)


if __name__ == "__main__":
    main()
