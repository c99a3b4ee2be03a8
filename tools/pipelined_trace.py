#!/usr/bin/env python3
"""Kernel durations and inter-kernel gaps of the headline launch in four cadences -- bare (the queue filled far ahead), launch +
fmd_demod_check_behind(2) / fmd_demod_check_prev / fmd_demod_check per step -- to be run under `rocprofv3 --kernel-trace` (tools/pipelined_gaps.py
reads the trace).  Prints the host's wall time per step for each cadence."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import rtl_sdr_rs_amd as fmd

dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
nch = 4096
bufs = []
for b in range(3):
    t = torch.empty((nch, bench.BLOCK), dtype=torch.uint8, device=dev)
    fmd.synth.fill_device(t.data_ptr(), nch, bench.BLOCK, sample_offset=b * (bench.BLOCK // 2), device_id=0, stream=stream)
    bufs.append(t)
cfg = fmd.DemodConfig(bench.FAST, bench.FAST, bench.SLOW, bench.D, 25)
bank = fmd.DemodBank(cfg, nch, device_id=0)
cap = bank.out_cap(bench.BLOCK)
outs = [torch.zeros((nch, cap), dtype=torch.int16, device=dev) for _ in range(3)]
launch = lambda i: bank.demodulate_device(bufs[i % 3].data_ptr(), bench.BLOCK, outs[i % 3].data_ptr(), cap, None, stream)
for i in range(300):
    launch(i)
bank.check()
res, guarded = {}, {}
for name, step in (("bare", lambda i: launch(i)), ("check_behind_2", lambda i: (launch(i), bank.check_behind(2))), ("check_prev", lambda i: (launch(i), bank.check_prev())),
                   ("check", lambda i: (launch(i), bank.check()))):
    torch.cuda.synchronize(); time.sleep(0.05)
    g0 = bank.f64_stats()["guarded"]
    t0 = time.perf_counter()
    for i in range(300):
        step(i)
    bank.check()
    res[name] = round((time.perf_counter() - t0) / 300 * 1e3, 4)
    guarded[name] = bank.f64_stats()["guarded"] - g0          # launches whose report buffer held a record: the light settle path
    time.sleep(0.05)
print(json.dumps({"host_wall_ms_per_step": res, "f64_guarded_samples": guarded, "order": ["warm-up 300", "bare 300", "check_behind(2) 300", "check_prev 300", "check 300"]}))
