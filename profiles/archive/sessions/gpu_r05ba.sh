#!/bin/bash
# session r05ba: the fused FIR kernel as shipped now (two digits: dense; one digit: sparse): all its tests, the FIR tests, both tap ranges timed
OUT=gpurun_out/r05ba; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_firdemod.py tests/test_fir.py tests/test_gpu_f64_guard.py tests/test_gpu_ref_kat.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest.log
python tools/ab_libs.py --firdemod --rounds 4 shipped= 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
python tools/ab_libs.py --firdemod --fir-taps-max 127 --rounds 4 shipped= 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
