#!/usr/bin/env python3
"""BASELINE configs[3] with the demodulator fused behind the FIR (fmd_firdemod_*): 127 taps, decimate 8, 256 channels x
2 MiB per call, 2.5 Msps -> 48 kHz.  One JSON line.  FMD_FD_KT / FMD_DBG / FMD_FD_NOREUSE (exp build) are tuning knobs;
BENCH_FD_DECIM=16 times the decimate-16 shape (1.25 Msps -> 48 kHz) instead."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rtl_sdr_rs_amd as fmd

nch, n, T, M, fast, slow = 256, 2 << 20, 127, 8, 2500000, 48000
if os.environ.get("BENCH_FD_DECIM"):
    M = int(os.environ["BENCH_FD_DECIM"]); fast = 20_000_000 // M
taps = np.random.default_rng(1).integers(-2047, 2048, T).astype(np.int16)
bank = fmd.FirDemodBank(taps, M, fast, slow, nch)
stream = torch.cuda.current_stream().cuda_stream
bufs = []
for b in range(3):
    t = torch.empty((nch, n), dtype=torch.uint8, device="cuda")
    fmd.synth.fill_device(t.data_ptr(), nch, n, sample_offset=b * (n // 2), stream=stream)
    bufs.append(t)
cap = bank.out_cap(n)
out = torch.zeros((nch, cap), dtype=torch.int16, device="cuda")
for i in range(150):
    bank.demodulate_device(bufs[i % 3].data_ptr(), n, out.data_ptr(), cap, stream)
torch.cuda.synchronize()
ts = []
for r in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(100):
        k = bank.demodulate_device(bufs[i % 3].data_ptr(), n, out.data_ptr(), cap, stream)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 100)
ms = sorted(ts)[1]
alg = nch * n + 2 * nch * k
knobs = {k: v for k, v in os.environ.items() if k.startswith("FMD_FD_") or k == "FMD_DBG"}
print(json.dumps({"decim": M, "knobs": knobs, "kernel": bank.kernel_name() if hasattr(bank, "kernel_name") else None, "ms": round(ms, 4),
                  "ms_all": [round(t, 4) for t in ts], "GBps": round(alg / ms / 1e6, 1), "frac": round(alg / ms / 1e6 / 8000, 4),
                  "tiling": bank.tiling(), "audio": k}))
