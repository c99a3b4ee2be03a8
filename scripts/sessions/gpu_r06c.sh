#!/bin/bash
# session r06c: the GPU suite on the tree after session r06b's findings (permuted image under the sparse FIR form only, with one address
# register per chunk; 16 rounds per wave at downsample 2; the equal-rate resampler pass as a 16-byte copy); same-process A/B against round 5.
OUT=gpurun_out/r06c; mkdir -p $OUT; export TMPDIR=/tmp
R5=rtl-sdr-rs_amd/libfmd_hip_r05.so
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -6 $OUT/pytest.txt
timeout 300 python tools/ab_libs.py --fir --rounds 4 r05=$R5 new= 2>/dev/null | tee $OUT/ab_fir.txt | cut -c1-260
timeout 300 python tools/ab_libs.py --fir --rounds 4 --out-bufs 4 r05=$R5 new= 2>/dev/null | tee $OUT/ab_fir_rot4.txt | cut -c1-260
timeout 300 python tools/ab_libs.py --fir --rounds 3 --fir-taps-max 127 r05=$R5 new= 2>/dev/null | tee $OUT/ab_fir8.txt | cut -c1-260
timeout 300 python tools/ab_libs.py --fir --rounds 3 --out-bufs 4 --fir-taps-max 127 r05=$R5 new= 2>/dev/null | tee $OUT/ab_fir8_rot4.txt | cut -c1-260
timeout 600 python tools/ab_libs.py --rounds 4 --cfg 1,48000,48000 --cfg 2,500000,32000 --cfg 2,96000,48000 --cfg 3,400000,48000 --cfg 4,256000,48000 --cfg 5,250000,44100 --cfg 6,170000,170000 --cfg ref --cfg 24 r05=$R5 new= 2>/dev/null | tee $OUT/ab_demod.txt | cut -c1-260
timeout 300 python - > $OUT/pipelined.json 2> $OUT/pipelined.err <<'PY'
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
out = torch.zeros((nch, cap), dtype=torch.int16, device=dev)
call = lambda i: bank.demodulate_device(bufs[i % 3].data_ptr(), bench.BLOCK, out.data_ptr(), cap, None, stream)
res = {}
for rep in range(3):
    ms, lo, hi, _ = bench.time_calls(torch, call, settle=100, steps=200, regions=3)
    bank.check()
    res.setdefault("bare_ms", []).append(round(ms, 4))
    res.setdefault("check_per_step", []).append(bench.extra_check_per_step(fmd, torch, bank, bufs, out, cap, stream)["ms_per_step"])
    res.setdefault("check_pipelined", []).append(bench.extra_check_pipelined(fmd, torch, bank, bufs, out, cap, stream))
print(json.dumps(res))
PY
cut -c1-900 $OUT/pipelined.json; tail -3 $OUT/pipelined.err
