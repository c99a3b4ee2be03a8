#!/bin/bash
# session r06e: (1) the bench line AFTER the committed PMC summary / bounds it quotes (profiles/r06_bench.json); (2) shader clock and
# socket power per configuration row on one box (tools/clock_probe.py: what the instruction-heavier rows' box spread follows);
# (3) where the last 3 % of the pipelined completion point go (GPU time by events beside the host's wall time; one / two output buffers).
OUT=gpurun_out/r06e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
for cfg in 24 ref 5,250000,44100 4,256000,48000 3,400000,48000 2,500000,32000 1,48000,48000 64,37500,8000; do
  timeout 120 python tools/clock_probe.py --cfg $cfg --seconds 2 shipped:0 2>/dev/null | grep -v '"variant": "idle"' >> $OUT/power_configs.jsonl
done
cut -c1-330 $OUT/power_configs.jsonl
timeout 300 python - > $OUT/pipelined_diag.json 2> $OUT/pipelined_diag.err <<'PY'
import json, sys, time
sys.path.insert(0, ".")
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
outs = [torch.zeros((nch, cap), dtype=torch.int16, device=dev) for _ in range(2)]
res = {}
def events(fn, steps=200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
    for i in range(steps):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / steps, 4), round((time.perf_counter() - t0) / steps * 1e3, 4)
for rep in range(3):
    for name, fn in (("bare_one_out", lambda i: bank.demodulate_device(bufs[i % 3].data_ptr(), bench.BLOCK, outs[0].data_ptr(), cap, None, stream)),
                     ("bare_two_outs", lambda i: bank.demodulate_device(bufs[i % 3].data_ptr(), bench.BLOCK, outs[i & 1].data_ptr(), cap, None, stream)),
                     ("pipelined_two_outs", lambda i: (bank.demodulate_device(bufs[i % 3].data_ptr(), bench.BLOCK, outs[i & 1].data_ptr(), cap, None, stream), bank.check_prev())),
                     ("pipelined_one_out", lambda i: (bank.demodulate_device(bufs[i % 3].data_ptr(), bench.BLOCK, outs[0].data_ptr(), cap, None, stream), bank.check_prev())),
                     ("check_every_step", lambda i: (bank.demodulate_device(bufs[i % 3].data_ptr(), bench.BLOCK, outs[0].data_ptr(), cap, None, stream), bank.check()))):
        for i in range(100):
            fn(i)
        bank.check()
        ev, wall = events(fn)
        bank.check()
        res.setdefault(name, []).append({"gpu_ms_by_events": ev, "host_wall_ms": wall})
print(json.dumps(res))
PY
cut -c1-1200 $OUT/pipelined_diag.json; tail -2 $OUT/pipelined_diag.err
