#!/usr/bin/env python3
"""BASELINE configs[1]: ONE synthetic FM channel at 2.4 Msps on one MI355X (device-resident input).
(a) DEFAULT_BUF_LENGTH per call -- launch-latency bound; (b) 64 MiB per call -- time-tiled
inside the channel; (c) the same 64 MiB handed over as 256 reference calls in one launch (fmd_demod_set_block_len).  Prints one JSON line; reported in DESIGN.md, not by bench.py."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtl_sdr_rs_amd as fmd

cfg = fmd.DemodConfig(240000, 240000, 32000, 10, 25)
res = {}
for name, n, steps, blk in [("262144_B_per_call", fmd.DEFAULT_BUF_LENGTH, 2000, 0), ("64_MiB_per_call", 64 << 20, 100, 0),
                            ("64_MiB_per_launch_as_256_reference_calls_of_262144_B", 64 << 20, 100, fmd.DEFAULT_BUF_LENGTH)]:
    bank = fmd.DemodBank(cfg, 1)
    bank.set_block_len(blk)
    iq = torch.empty((1, n), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    fmd.synth.fill_device(iq.data_ptr(), 1, n, stream=stream)
    cap = bank.out_cap(n)
    out = torch.zeros((1, cap), dtype=torch.int16, device="cuda")
    for _ in range(50):
        bank.demodulate_device(iq.data_ptr(), n, out.data_ptr(), cap, None, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        bank.demodulate_device(iq.data_ptr(), n, out.data_ptr(), cap, None, stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    res[name] = {"ms_per_call": round(ms, 5), "iq_msamples_per_s": round(n / 2 / ms / 1e3, 1),
                 "realtime_factor_at_2.4Msps": round(n / 2 / ms / 1e3 / 2.4, 1), "GBps": round(n / ms / 1e6, 1)}
print(json.dumps({"workload": "BASELINE configs[1]: 1 FM channel @ 2.4 Msps, device-resident input", **res}))
