#!/bin/bash
# session r05b: ADVICE r4 fixes on the GPU (checkpoint v2, kernel names, event-based stream ordering) + what the event record per
# launch costs: the round-4 library against this one, alternating processes, headline + cfg-ref + check_per_step
OUT=gpurun_out/r05b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_firdemod.py tests/test_gpu_boundary.py tests/test_gpu_ref_kat.py tests/test_fir.py tests/test_gpu_sink.py tests/test_c_abi.py -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest.log
for r in 1 2 3; do
  for lib in r04 new; do
    if [ $lib = r04 ]; then export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_r04.so; else unset FMD_LIB; fi
    python tools/ab.py --rounds 2 --cfg 24 --cfg ref $lib: 2>/dev/null | grep '^{"cfg"' | sed "s/^/$r /" >> $OUT/ab_event.txt
  done
done
unset FMD_LIB
cat $OUT/ab_event.txt | cut -c1-150
python bench.py --no-cpu 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['frac'], json.dumps(r['extra'].get('check_per_step')), json.dumps(r['extra'].get('cfg_ref'))[:300])" | tee $OUT/bench_short.txt
