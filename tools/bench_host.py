#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (fmd_demod_demodulate_batch: H2D + kernel + D2H per
call, pageable host memory).  Reported in DESIGN.md only -- never bench.py's `value`."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtl_sdr_rs_amd as fmd

nch, N = 1024, fmd.DEFAULT_BUF_LENGTH
cfg = fmd.DemodConfig(240000, 240000, 32000, 10, 25)
bank = fmd.DemodBank(cfg, nch)
iq = fmd.synth.synth_iq(8, N)
iq = np.ascontiguousarray(np.tile(iq, (nch // 8, 1)))
for _ in range(3):
    bank.demodulate_batch(iq)
t0 = time.perf_counter(); steps = 10
for _ in range(steps):
    bank.demodulate_batch(iq)
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"workload": "%d channels x %d B per call, host buffers (pageable), H2D + kernel + D2H + Python list copy" % (nch, N),
                  "ms_per_call": round(dt * 1e3, 3), "iq_msamples_per_s": round(nch * N / 2 / dt / 1e6, 1),
                  "host_GBps": round(nch * N / dt / 1e9, 2)}))
