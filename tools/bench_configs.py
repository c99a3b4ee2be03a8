#!/usr/bin/env python3
"""Throughput of the demod path across the supported domain (not the headline bench): 4096 channels x 262144 B,
device-resident input, for several (downsample, rate_out, rate_resample).  One JSON line per configuration."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtl_sdr_rs_amd as fmd

CONFIGS = [("cfg-ref (optimal_settings(94.9 MHz, 170 kHz), simple_fm.rs:25-27)", 6, 170000, 32000),
           ("cfg-2.4 (BASELINE configs[2])", 10, 240000, 32000),
           ("D=4 256k->48k", 4, 256000, 48000), ("D=8 250k->44.1k", 8, 250000, 44100), ("D=2 500k->32k", 2, 500000, 32000),
           ("D=3 (odd) 400k->48k", 3, 400000, 48000), ("D=7 (odd) 166666->32k", 7, 166666, 32000), ("D=5 (odd) 250k->44.1k", 5, 250000, 44100),
           ("D=1 48k->48k", 1, 48000, 48000), ("D=16 150k->32k", 16, 150000, 32000), ("D=64 37.5k->8k", 64, 37500, 8000),
           ("D=32 512k->32k", 32, 512000, 32000), ("D=12 192k->32k", 12, 192000, 32000), ("D=13 (odd) 208k->32k", 13, 208000, 32000), ("D=14 224k->32k", 14, 224000, 32000)]
if len(sys.argv) > 1:
    CONFIGS = [c for c in CONFIGS if any(a in c[0] for a in sys.argv[1:])]
nch, N = 4096, fmd.DEFAULT_BUF_LENGTH
stream = torch.cuda.current_stream().cuda_stream
bufs = []
for b in range(3):
    t = torch.empty((nch, N), dtype=torch.uint8, device="cuda")
    fmd.synth.fill_device(t.data_ptr(), nch, N, sample_offset=b * (N // 2), stream=stream)
    bufs.append(t)
for name, D, fast, slow in CONFIGS:
    cfg = fmd.DemodConfig(fast, fast, slow, D, max(1, (1 << 15) // (128 * D)))
    bank = fmd.DemodBank(cfg, nch)
    cap = bank.out_cap(N)
    out = torch.zeros((nch, cap), dtype=torch.int16, device="cuda")
    for i in range(150):
        bank.demodulate_device(bufs[i % 3].data_ptr(), N, out.data_ptr(), cap, None, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    steps = 100
    e0.record()
    for i in range(steps):
        bank.demodulate_device(bufs[i % 3].data_ptr(), N, out.data_ptr(), cap, None, stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    alg = nch * N + 2 * int(bank.last_out_len().sum())
    print(json.dumps({"config": name, "cfg": [D, fast, slow], "kernel": bank.last_kernel(), "ms_per_call": round(ms, 4), "iq_msamples_per_s": round(nch * N / 2 / ms / 1e3, 0),
                      "algorithmic_GBps": round(alg / ms / 1e6, 1), "hbm_frac_of_8TBps": round(alg / ms / 1e6 / 8000, 4),
                      "tiling": bank.tiling()}), flush=True)
    bank.close(); del out
