#!/bin/bash
# session r06f: the GPU suite with the opt-in event ordering (fmd_demod_set_event_ordering) and the FIR kernel-name test; what the
# opt-in costs per launch (same process, same handle configuration, headline and cfg-ref).
OUT=gpurun_out/r06f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -6 $OUT/pytest.txt
timeout 300 python - > $OUT/event_mode.json 2> $OUT/event_mode.err <<'PY'
import json, sys
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
res = {}
for name, (D, fast, slow) in (("headline", (bench.D, bench.FAST, bench.SLOW)), ("cfg_ref", bench.CFG_REF)):
    cfg = fmd.DemodConfig(fast, fast, slow, D, max(1, (1 << 15) // (128 * D)))
    banks = {"default": fmd.DemodBank(cfg, nch, device_id=0), "event_ordering": fmd.DemodBank(cfg, nch, device_id=0)}
    banks["event_ordering"].set_event_ordering(True)
    cap = banks["default"].out_cap(bench.BLOCK)
    out = torch.zeros((nch, cap), dtype=torch.int16, device=dev)
    for rnd in range(4):
        for k, bank in banks.items():
            ms, lo, hi, _ = bench.time_calls(torch, lambda i: bank.demodulate_device(bufs[i % 3].data_ptr(), bench.BLOCK, out.data_ptr(), cap, None, stream), settle=100, steps=100, regions=3)
            bank.check()
            res.setdefault(name, {}).setdefault(k, []).append(round(ms, 4))
print(json.dumps(res))
PY
cat $OUT/event_mode.json; tail -2 $OUT/event_mode.err
