#!/bin/bash
# session r06i: the completion point two launches back (fmd_demod_check_behind: ring of three) -- its GPU tests (forced patches with two
# and three launches in flight, the replay chain), the whole suite, and the kernel trace of the four cadences.
OUT=gpurun_out/r06i; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_f64_guard.py tests/test_gpu_boundary.py -m gpu -x -q > $OUT/pytest_f64.txt 2>&1; tail -8 $OUT/pytest_f64.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -6 $OUT/pytest.txt
rm -rf $OUT/pt
timeout 300 rocprofv3 --kernel-trace -d $OUT/pt -o pt -f csv -- python3 tools/pipelined_trace.py > $OUT/pipelined_trace.json 2> $OUT/pt.err
python3 tools/pipelined_gaps.py $OUT/pt > $OUT/pipelined_gaps.json; cat $OUT/pipelined_trace.json $OUT/pipelined_gaps.json
rm -rf $OUT/pt
timeout 300 python - > $OUT/pipelined_plain.json 2> $OUT/pipelined_plain.err <<'PY'
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
cfg = fmd.DemodConfig(bench.FAST, bench.FAST, bench.SLOW, bench.D, 25)
bank = fmd.DemodBank(cfg, nch, device_id=0)
cap = bank.out_cap(bench.BLOCK)
out = torch.zeros((nch, cap), dtype=torch.int16, device=dev)
res = {"bare_ms": [], "check_per_step": [], "check_pipelined": []}
for rep in range(4):
    ms, lo, hi, _ = bench.time_calls(torch, lambda i: bank.demodulate_device(bufs[i % 3].data_ptr(), bench.BLOCK, out.data_ptr(), cap, None, stream), settle=100, steps=200, regions=3)
    bank.check()
    res["bare_ms"].append(round(ms, 4))
    res["check_per_step"].append(bench.extra_check_per_step(fmd, torch, bank, bufs, out, cap, stream)["ms_per_step"])
    res["check_pipelined"].append(bench.extra_check_pipelined(fmd, torch, bank, bufs, out, cap, stream))
print(json.dumps(res))
PY
python3 -c "
import json; d=json.load(open('$OUT/pipelined_plain.json'))
print(d['bare_ms'], d['check_per_step']); [print(x['ms_per_step'], x['one_launch_back']['ms_per_step'], x['host_ms_in_enqueue']) for x in d['check_pipelined']]"; tail -2 $OUT/pipelined_plain.err
