#!/usr/bin/env python3
"""Interleaved A/B of library BUILDS inside ONE process on ONE box: every build is dlopen()ed side by side (its own handle, its own
code objects), all run on the same input buffers and the same clocks, one after the other, round after round -- processes of
their own per build (scripts/ab_libs.sh) turned out to differ by up to 2 % between themselves whatever they load
(profiles/r05_experiments.md section 2).  Usage:
    tools/ab_libs.py [--cfg ref|24|D,fast,slow]... [--rounds 5] [--steps 100] name=path/to/lib.so name2=...   ('name=' : the shipped library)
    name=path@VAR=val,VAR2=val : environment variables set while THAT build's handle is created (knobs of the -DFMD_EXPERIMENT build)
Prints one JSON line per (config, build) with the per-round times, the median and the difference to the first build."""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rtl_sdr_rs_amd as fmd
from rtl_sdr_rs_amd import _ffi

NAMED = {"ref": (6, 170000, 32000), "24": (10, 240000, 32000)}


def parse_build(b):
    name, _, rest = b.partition("=")
    path, _, envs = rest.partition("@")
    env = dict(e.split("=", 1) for e in envs.split(",") if e)
    return name, load(os.path.join(ROOT, path) if path else _ffi.SO_PATH), env


class knobs:
    """environment of one build while its handle is created"""
    def __init__(self, env):
        self.env, self.saved = env, {}
    def __enter__(self):
        for k, v in self.env.items():
            self.saved[k] = os.environ.get(k); os.environ[k] = v
    def __exit__(self, *a):
        for k, v in self.saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


_loaded = {}
def load(path):
    if path in _loaded:
        return _loaded[path]
    l = _loaded[path] = C.CDLL(path)
    for name, (res, args) in _ffi.PROTOTYPES.items():
        try:
            fn = getattr(l, name)
        except AttributeError:
            continue
        fn.restype, fn.argtypes = res, args
    return l


def firdemod(a):
    import numpy as np
    parsed = [parse_build(b) for b in a.builds]
    builds = [(n_, l_) for n_, l_, _ in parsed]
    envs = [e_ for _, _, e_ in parsed]
    nch, n, T, M, fast, slow = 256, 2 << 20, 127, 8, a.fd_fast, a.fd_slow
    taps = np.random.default_rng(1).integers(-a.fir_taps_max, a.fir_taps_max + 1, T).astype(np.int16)
    shift = fmd.auto_shift(taps)
    stream = torch.cuda.current_stream().cuda_stream
    bufs = []
    for b in range(3):
        t = torch.empty((nch, n), dtype=torch.uint8, device="cuda")
        fmd.synth.fill_device(t.data_ptr(), nch, n, sample_offset=b * (n // 2), stream=stream)
        bufs.append(t)
    cap = int(builds[0][1].fmd_firdemod_out_cap(M, fast, slow, n))
    out = torch.zeros((nch, cap), dtype=torch.int16, device="cuda")
    hs = []
    for (name, l), env in zip(builds, envs):
        h = C.c_void_p()
        dev = fmd.DeviceConfig(nch, -1, 0)
        with knobs(env):
            rc = l.fmd_firdemod_new(taps.ctypes.data_as(C.POINTER(C.c_int16)), T, M, shift, fast, slow, C.byref(dev), C.byref(h))
        assert rc == 0, (name, rc)
        hs.append(h)
    res = {name: [] for name, _ in builds}
    k = C.c_size_t(0)
    for rnd in range(a.rounds):
        for (name, l), h in zip(builds, hs):
            for i in range(a.settle):
                l.fmd_firdemod_demodulate_device(h, bufs[i % 3].data_ptr(), n, out.data_ptr(), cap, C.byref(k), stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.steps):
                l.fmd_firdemod_demodulate_device(h, bufs[i % 3].data_ptr(), n, out.data_ptr(), cap, C.byref(k), stream)
            e1.record(); torch.cuda.synchronize()
            assert l.fmd_firdemod_check(h) == 0
            res[name].append(e0.elapsed_time(e1) / a.steps)
    alg = nch * n + 2 * nch * k.value
    base = None
    for (name, l), h in zip(builds, hs):
        ts = sorted(res[name]); med = ts[len(ts) // 2]
        base = base or med
        print(json.dumps({"cfg": "fused FIR 127 / 8, %d -> %d Hz, |tap| <= %d" % (fast, slow, a.fir_taps_max), "build": name, "ms": [round(t, 4) for t in res[name]], "median_ms": round(med, 4),
                          "frac": round(alg / med / 1e6 / 8000, 4), "vs_first_pct": round(100 * (med / base - 1), 2)}), flush=True)
        l.fmd_firdemod_free(h)


def fir(a):
    """BASELINE configs[3] through the stand-alone matrix-core FIR (127 taps / 8, 256 channels x 2 MiB; --fir-taps-max for the one-digit form)"""
    import numpy as np
    parsed = [parse_build(b) for b in a.builds]
    builds = [(n_, l_) for n_, l_, _ in parsed]
    envs = [e_ for _, _, e_ in parsed]
    nch, n, T, M = a.fir_channels, a.fir_bytes, 127, 8
    taps = np.random.default_rng(1).integers(-a.fir_taps_max, a.fir_taps_max + 1, T).astype(np.int16)
    stream = torch.cuda.current_stream().cuda_stream
    bufs = []
    for b in range(3):
        t = torch.empty((nch, n), dtype=torch.uint8, device="cuda")
        fmd.synth.fill_device(t.data_ptr(), nch, n, sample_offset=b * (n // 2), stream=stream)
        bufs.append(t)
    cap = int(builds[0][1].fmd_fir_out_cap(T, M, n))
    outs = [torch.zeros((nch, cap, 2), dtype=torch.int32, device="cuda") for _ in range(a.out_bufs)]
    hs = []
    for (name, l), env in zip(builds, envs):
        h = C.c_void_p()
        dev = fmd.DeviceConfig(nch, -1, 0)
        with knobs(env):
            rc = l.fmd_fir_new(taps.ctypes.data_as(C.POINTER(C.c_int16)), T, M, C.byref(dev), C.byref(h))
        assert rc == 0, (name, rc)
        hs.append(h)
    res = {name: [] for name, _ in builds}
    k = C.c_size_t(0)
    for rnd in range(a.rounds):
        for (name, l), h in zip(builds, hs):
            for i in range(a.settle):
                l.fmd_fir_filter_device(h, bufs[i % 3].data_ptr(), n, outs[i % a.out_bufs].data_ptr(), cap, C.byref(k), stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.steps):
                l.fmd_fir_filter_device(h, bufs[i % 3].data_ptr(), n, outs[i % a.out_bufs].data_ptr(), cap, C.byref(k), stream)
            e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / a.steps)
    alg = nch * n + 8 * nch * k.value
    base = None
    for (name, l), h in zip(builds, hs):
        ts = sorted(res[name]); med = ts[len(ts) // 2]
        base = base or med
        print(json.dumps({"cfg": "config4 FIR, %d x %d B, |tap| <= %d" % (nch, n, a.fir_taps_max), "build": name, "ms": [round(t, 4) for t in res[name]], "median_ms": round(med, 4),
                          "frac": round(alg / med / 1e6 / 8000, 4), "vs_first_pct": round(100 * (med / base - 1), 2)}), flush=True)
        l.fmd_fir_free(h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", action="append", default=[])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--settle", type=int, default=100)
    ap.add_argument("--channels", type=int, default=4096)
    ap.add_argument("--cap-align", type=int, default=1, help="round the output row length (samples) up to a multiple of this")
    ap.add_argument("--firdemod", action="store_true", help="BASELINE configs[3] through the fused FIR kernel (127 taps, decimate 8, 256 channels x 2 MiB) instead of the demodulation kernels")
    ap.add_argument("--fir", action="store_true", help="BASELINE configs[3] through the stand-alone FIR kernel")
    ap.add_argument("--fir-taps-max", type=int, default=2047)
    ap.add_argument("--fir-bytes", type=int, default=2 << 20)
    ap.add_argument("--fd-fast", type=int, default=2500000, help="--firdemod: rate after the FIR")
    ap.add_argument("--fd-slow", type=int, default=48000, help="--firdemod: audio rate")
    ap.add_argument("--out-bufs", type=int, default=1, help="rotate the calls over this many output buffers (more than the 256 MB memory-side cache holds: every written byte goes to HBM)")
    ap.add_argument("--fir-channels", type=int, default=256)
    ap.add_argument("builds", nargs="+")
    a = ap.parse_args()
    if a.firdemod:
        return firdemod(a)
    if a.fir:
        return fir(a)
    cfgs = [NAMED[c] if c in NAMED else tuple(int(x) for x in c.split(",")) for c in (a.cfg or ["24", "ref"])]
    parsed = [parse_build(b) for b in a.builds]
    builds = [(n_, l_) for n_, l_, _ in parsed]
    envs = [e_ for _, _, e_ in parsed]
    nch, N = a.channels, fmd.DEFAULT_BUF_LENGTH
    stream = torch.cuda.current_stream().cuda_stream
    bufs = []
    for b in range(3):
        t = torch.empty((nch, N), dtype=torch.uint8, device="cuda")
        fmd.synth.fill_device(t.data_ptr(), nch, N, sample_offset=b * (N // 2), stream=stream)
        bufs.append(t)
    torch.cuda.synchronize()
    for D, fast, slow in cfgs:
        cfg = fmd.DemodConfig(fast, fast, slow, D, max(1, (1 << 15) // (128 * D)))
        cap = int(builds[0][1].fmd_out_cap(C.byref(cfg), N))
        cap = -(-cap // a.cap_align) * a.cap_align
        outs = [torch.zeros((nch, cap), dtype=torch.int16, device="cuda") for _ in range(a.out_bufs)]
        hs = []
        for (name, l), env in zip(builds, envs):
            h = C.c_void_p()
            dev = fmd.DeviceConfig(nch, -1, 0)
            with knobs(env):
                rc = l.fmd_demod_new(C.byref(cfg), C.byref(dev), C.byref(h))
            assert rc == 0, (name, rc)
            hs.append(h)
        res = {name: [] for name, _ in builds}
        kern = {}
        for rnd in range(a.rounds):
            for (name, l), h in zip(builds, hs):
                for i in range(a.settle):
                    l.fmd_demod_demodulate_device(h, bufs[i % 3].data_ptr(), N, outs[i % a.out_bufs].data_ptr(), cap, None, stream)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(a.steps):
                    l.fmd_demod_demodulate_device(h, bufs[i % 3].data_ptr(), N, outs[i % a.out_bufs].data_ptr(), cap, None, stream)
                e1.record(); torch.cuda.synchronize()
                assert l.fmd_demod_check(h) == 0
                res[name].append(e0.elapsed_time(e1) / a.steps)
                buf = C.create_string_buffer(128)
                l.fmd_demod_last_kernel(h, buf, len(buf))
                t1, t2, t3 = C.c_uint32(), C.c_uint32(), C.c_uint32()
                l.fmd_demod_tiling(h, C.byref(t1), C.byref(t2), C.byref(t3))
                lens = (C.c_size_t * nch)()
                l.fmd_demod_last_out_len(h, lens)
                kern[name] = "%s kt=%d lds=%d tiles=%d" % (buf.value.decode(), t1.value, t2.value, -(-int(lens[0]) // max(1, t1.value)))
        base = None
        for (name, l), h in zip(builds, hs):
            ts = sorted(res[name]); med = ts[len(ts) // 2]
            base = base or med
            print(json.dumps({"cfg": [D, fast, slow], "build": name, "kernel": kern[name], "ms": [round(t, 4) for t in res[name]], "median_ms": round(med, 4),
                              "vs_first_pct": round(100 * (med / base - 1), 2)}), flush=True)
            l.fmd_demod_free(h)
        del outs


if __name__ == "__main__":
    main()
