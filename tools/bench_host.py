#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (fmd_demod_demodulate_batch: H2D + kernel + D2H per
call), with pageable numpy buffers and with page-locked buffers from fmd_host_alloc.  Reported in DESIGN.md
only -- never bench.py's `value`."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtl_sdr_rs_amd as fmd

nch, N = 1024, fmd.DEFAULT_BUF_LENGTH
cfg = fmd.DemodConfig(240000, 240000, 32000, 10, 25)
bank = fmd.DemodBank(cfg, nch)
cap = bank.out_cap(N)
src = np.ascontiguousarray(np.tile(fmd.synth.synth_iq(8, N), (nch // 8, 1)))
res = {"workload": "%d channels x %d B per call, host buffers: H2D + kernel + D2H" % (nch, N)}


def run(iq, out, steps=10):
    for _ in range(3):
        bank.demodulate_batch_into(iq, out)
    t0 = time.perf_counter()
    for _ in range(steps):
        bank.demodulate_batch_into(iq, out)
    dt = (time.perf_counter() - t0) / steps
    return {"ms_per_call": round(dt * 1e3, 3), "iq_msamples_per_s": round(nch * N / 2 / dt / 1e6, 1),
            "host_GBps": round(nch * N / dt / 1e9, 2)}


res["pageable"] = run(src, np.empty((nch, cap), np.int16))
pin_in, pin_out = fmd.PinnedBuffer((nch, N), np.uint8), fmd.PinnedBuffer((nch, cap), np.int16)
pin_in.array[:] = src
res["pinned"] = run(pin_in.array, pin_out.array)
print(json.dumps(res))
