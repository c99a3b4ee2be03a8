#!/usr/bin/env python3
"""What fmd_demod_check costs per step at the headline configuration, in pieces: bare launches; launch + a stream synchronisation (no
report read at all: the floor of any completion point on the newest launch); launch + fmd_demod_check.  Host wall time, median of 5 x 300."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import rtl_sdr_rs_amd as fmd

dev = torch.device("cuda", 0); ts = torch.cuda.current_stream(); stream = ts.cuda_stream
nch = 4096
bufs = []
for b in range(3):
    t = torch.empty((nch, bench.BLOCK), dtype=torch.uint8, device=dev)
    fmd.synth.fill_device(t.data_ptr(), nch, bench.BLOCK, sample_offset=b * (bench.BLOCK // 2), device_id=0, stream=stream)
    bufs.append(t)
cfg = fmd.DemodConfig(bench.FAST, bench.FAST, bench.SLOW, bench.D, 25)
bank = fmd.DemodBank(cfg, nch, device_id=0)
cap = bank.out_cap(bench.BLOCK)
out = torch.zeros((nch, cap), dtype=torch.int16, device=dev)
launch = lambda i: bank.demodulate_device(bufs[i % 3].data_ptr(), bench.BLOCK, out.data_ptr(), cap, None, stream)
for i in range(300):
    launch(i)
bank.check()
res = {}
def spin(i):
    launch(i)
    while not ts.query():
        pass

for name, step in (("bare", lambda i: launch(i)), ("stream_sync", lambda i: (launch(i), ts.synchronize())), ("stream_query_spin", spin), ("check", lambda i: (launch(i), bank.check()))):
    runs = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(300):
            step(i)
        bank.check()
        runs.append((time.perf_counter() - t0) / 300 * 1e3)
    res[name] = round(sorted(runs)[2], 4)
print(json.dumps({"host_wall_ms_per_step": res, "f64": bank.f64_stats()}))
