#!/bin/bash
# session r05t: the fused FIR kernel with its post-barrier kernel arguments prefetched under the DMAs (opaque copies): parity + A/B
OUT=gpurun_out/r05t; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_firdemod.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
python tools/ab_libs.py --firdemod --rounds 6 s4=rtl-sdr-rs_amd/libfmd_hip_s4.so new= s4b=rtl-sdr-rs_amd/libfmd_hip_s4.so newb= 2>/dev/null | tee $OUT/ab_fd.jsonl | cut -c1-200
