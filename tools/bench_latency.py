#!/usr/bin/env python3
"""Host wall time of ONE reference-shaped call -- Demod::demodulate on one read_sync buffer (simple_fm.rs:135-156): page-locked
host buffers, H2D + kernel + D2H + completion, one channel -- and of the device entry + fmd_demod_check.  Microseconds per call."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtl_sdr_rs_amd as fmd

N = fmd.DEFAULT_BUF_LENGTH
res = {}
for name, cfg in (("cfg-2.4 (downsample 10, 240 k -> 32 k)", fmd.DemodConfig(240000, 240000, 32000, 10, 25)),
                  ("cfg-ref (downsample 6, 170 k -> 32 k)", fmd.DemodConfig(170000, 170000, 32000, 6, 25))):
    bank = fmd.DemodBank(cfg, 1)
    cap = bank.out_cap(N)
    pin_in, pin_out = fmd.PinnedBuffer((1, N), np.uint8), fmd.PinnedBuffer((1, cap), np.int16)
    pin_in.array[:] = fmd.synth.synth_iq(1, N)
    for _ in range(200):
        bank.demodulate_batch_into(pin_in.array, pin_out.array)
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(500):
            bank.demodulate_batch_into(pin_in.array, pin_out.array)
        best = min(best, (time.perf_counter() - t0) / 500)
    res[name] = {"host_call_us": round(best * 1e6, 1), "realtime_factor_at_2.4Msps": round((N / 2 / 2.4e6) / best, 1)}
    bank.close()
print(json.dumps(res))
