#!/usr/bin/env python3
"""Timeline of ONE launch of the demodulation tile kernel from the probe of the -DFMD_EXPERIMENT build (ablation bit 29,
csrc/fmd_tile_body.h): every wave records its entry, the end of its staging barrier and its end (shader-clock counter) together
with the hardware slot it ran in.  Answers what the counters cannot: how many tiles a CU really holds over the launch, how much
of a tile's life is staging and how much compute, and how long a freed slot stays empty before the next block runs in it.
Usage: tools/timeline.py [--cfg ref|24|D,fast,slow]... [--lib rtl-sdr-rs_amd/libfmd_hip_exp.so] [--dbg extra ablation bits]"""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import rtl_sdr_rs_amd as fmd
from rtl_sdr_rs_amd import _ffi

NAMED = {"ref": (6, 170000, 32000), "24": (10, 240000, 32000)}


def load(path):
    l = C.CDLL(path)
    for name, (res, args) in _ffi.PROTOTYPES.items():
        try:
            fn = getattr(l, name)
        except AttributeError:
            continue
        fn.restype, fn.argtypes = res, args
    return l


def analyse(rec, nch_pad):
    """rec: [n_waves][8] uint32 (t0 lo, t0 hi, t1 - t0, t2 - t0, HW_ID, XCC_ID, channel, tile)"""
    rec = rec[rec[:, 3] != 0]
    t0 = rec[:, 0].astype(np.uint64) | (rec[:, 1].astype(np.uint64) << np.uint64(32))
    d1, d2 = rec[:, 2].astype(np.int64), rec[:, 3].astype(np.int64)
    hw, xcc = rec[:, 4], rec[:, 5] & 0xF
    cu = (xcc.astype(np.int64) << 12) | ((hw >> 8) & 0xFF).astype(np.int64)          # XCC, SE, SH, CU
    simd = (cu << 2) | ((hw >> 4) & 3)
    blk = rec[:, 6].astype(np.int64) * 65536 + (rec[:, 7] >> 16).astype(np.int64)     # (channel, block of the channel-call)
    out = {"tile_waves": int(len(rec)), "cus": int(len(np.unique(cu))), "simds": int(len(np.unique(simd)))}
    out["tile_life_clk"] = {"mean": float(d2.mean()), "p10": float(np.percentile(d2, 10)), "p90": float(np.percentile(d2, 90))}
    out["staging_clk"] = {"mean": float(d1.mean()), "p10": float(np.percentile(d1, 10)), "p90": float(np.percentile(d1, 90))}
    out["compute_clk"] = {"mean": float((d2 - d1).mean()), "p10": float(np.percentile(d2 - d1, 10)), "p90": float(np.percentile(d2 - d1, 90))}
    # per CU: blocks resident over time (a block = its four waves over all its tiles: first entry ... last end)
    order = np.argsort(blk, kind="stable")
    b_ids, first = np.unique(blk[order], return_index=True)
    b_t0 = np.minimum.reduceat(t0[order], first).astype(np.int64)
    b_t2 = np.maximum.reduceat((t0 + d2.astype(np.uint64))[order], first).astype(np.int64)
    b_stage = np.add.reduceat(d1[order], first) / 4.0        # clocks the block spent staging (mean over its four waves)
    b_cu = cu[order][first]
    res, stg, spans, gaps = [], [], [], []
    for c in np.unique(b_cu):
        m = b_cu == c
        s, e = b_t0[m], b_t2[m]
        span = e.max() - s.min()
        spans.append(span)
        res.append((e - s).sum() / span)                    # average number of resident blocks
        stg.append(b_stage[m].sum() / span)                 # ... of which in a staging phase
        # slot turn-over: the k-th block to START on a CU takes the slot of the (k - n)-th block to END, n = the most the CU ever held
        ends = np.sort(e); starts = np.sort(s)
        ev = np.concatenate([np.stack([s, np.ones_like(s)], 1), np.stack([e, -np.ones_like(e)], 1)])
        ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]             # (ends before starts at equal times)
        n_res = int(np.cumsum(ev[:, 1]).max())
        k = min(len(starts) - n_res, len(ends))
        if k > 0:
            gaps.append(np.maximum(starts[n_res:n_res + k] - ends[:k], 0))
    gaps = np.concatenate(gaps) if gaps else np.zeros(1)
    out["resident_blocks_per_cu"] = {"mean": float(np.mean(res)), "min": float(np.min(res)), "max": float(np.max(res))}
    out["staging_blocks_per_cu"] = {"mean": float(np.mean(stg))}
    out["cu_span_clk"] = {"mean": float(np.mean(spans)), "min": float(np.min(spans)), "max": float(np.max(spans))}
    out["slot_turnover_gap_clk"] = {"mean": float(gaps.mean()), "p50": float(np.percentile(gaps, 50)), "p90": float(np.percentile(gaps, 90))}
    out["blocks"] = int(len(b_ids))
    out["max_resident_blocks_last_cu"] = n_res
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", action="append", default=[])
    ap.add_argument("--lib", default="rtl-sdr-rs_amd/libfmd_hip_exp.so")
    ap.add_argument("--dbg", type=int, default=0)
    ap.add_argument("--kt", type=int, default=0, help="FMD_KT of the experiment build: audio samples per tile")
    ap.add_argument("--channels", type=int, default=4096)
    ap.add_argument("--dump", default=None, help="write the raw records of the last configuration to this .npy file")
    a = ap.parse_args()
    cfgs = [NAMED[c] if c in NAMED else tuple(int(x) for x in c.split(",")) for c in (a.cfg or ["24", "ref"])]
    l = load(os.path.join(ROOT, a.lib))
    nch, N = a.channels, fmd.DEFAULT_BUF_LENGTH
    stream = torch.cuda.current_stream().cuda_stream
    bufs = []
    for b in range(3):
        t = torch.empty((nch, N), dtype=torch.uint8, device="cuda")
        fmd.synth.fill_device(t.data_ptr(), nch, N, sample_offset=b * (N // 2), stream=stream)
        bufs.append(t)
    nch_pad = (nch + 1023) & ~1023
    max_waves = 4 * nch * 64
    trace = torch.zeros(nch_pad + 8 * max_waves, dtype=torch.int32, device="cuda")
    for D, fast, slow in cfgs:
        cfg = fmd.DemodConfig(fast, fast, slow, D, max(1, (1 << 15) // (128 * D)))
        cap = int(l.fmd_out_cap(C.byref(cfg), N))
        out = torch.zeros((nch, cap), dtype=torch.int16, device="cuda")
        for dbg in (a.dbg, a.dbg | (1 << 29)):
            os.environ["FMD_DBG"] = str(dbg)
            if a.kt: os.environ["FMD_KT"] = str(a.kt)
            h = C.c_void_p()
            dev = fmd.DeviceConfig(nch, -1, 0)
            assert l.fmd_demod_new(C.byref(cfg), C.byref(dev), C.byref(h)) == 0
            for i in range(100):
                l.fmd_demod_demodulate_device(h, bufs[i % 3].data_ptr(), N, out.data_ptr(), cap, None, stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(100):
                l.fmd_demod_demodulate_device(h, bufs[i % 3].data_ptr(), N, out.data_ptr(), cap, trace.data_ptr() if dbg >> 29 else None, stream)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 100
            if dbg >> 29:
                trace.zero_()
                l.fmd_demod_demodulate_device(h, bufs[0].data_ptr(), N, out.data_ptr(), cap, trace.data_ptr(), stream)
                torch.cuda.synchronize()
                rec = trace[nch_pad:].cpu().numpy().view(np.uint32).reshape(-1, 8)
                res = analyse(rec, nch_pad)
                res.update({"cfg": [D, fast, slow], "ms_per_call_with_probe": round(ms, 4), "ms_per_call_without": round(ms_plain, 4)})
                buf = C.create_string_buffer(128)
                l.fmd_demod_last_kernel(h, buf, len(buf))
                res["kernel"] = buf.value.decode()
                t1, t2, t3 = C.c_uint32(), C.c_uint32(), C.c_uint32()
                l.fmd_demod_tiling(h, C.byref(t1), C.byref(t2), C.byref(t3))
                res["audio_per_tile"], res["lds_bytes"] = t1.value, t2.value
                print(json.dumps(res), flush=True)
                if a.dump:
                    np.save(a.dump, rec[rec[:, 3] != 0])
            else:
                ms_plain = ms
            assert l.fmd_demod_check(h) == 0
            l.fmd_demod_free(h)
        del out


if __name__ == "__main__":
    main()
