#!/bin/bash
# session r05az: sparse against dense matrix phase of the fused FIR kernel, shipped-flavour builds side by side; the variants test again
OUT=gpurun_out/r05az; mkdir -p $OUT; export TMPDIR=/tmp
P=rtl-sdr-rs_amd
python tools/ab_libs.py --firdemod --rounds 5 sparse= dense=$P/libfmd_hip_dn.so 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
python tools/ab_libs.py --firdemod --rounds 5 dense=$P/libfmd_hip_dn.so sparse= 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
python tools/ab_libs.py --firdemod --fir-taps-max 127 --rounds 5 sparse= dense=$P/libfmd_hip_dn.so 2>/dev/null | tee -a $OUT/ab.txt | cut -c1-220
timeout 1200 python -m pytest tests/test_firdemod.py -x -q -m gpu -k "variants" 2>&1 | tail -4 | tee $OUT/pytest.log
