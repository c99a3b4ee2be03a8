#!/bin/bash
# session r04u: fused FIR kernel: additive constants in the C operand of the first matrix instruction, group sums by bit masks
# instead of VCC selects: parity + A/B (three alternating rounds of tools/bench_firdemod.py)
OUT=gpurun_out/r04u; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_firdemod.py tests/test_fir.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
for i in 1 2 3; do
  for v in base new; do
    L=libfmd_hip.so; [ $v = base ] && L=libfmd_hip_base.so
    echo "$v $(FMD_LIB=$PWD/rtl-sdr-rs_amd/$L python3 tools/bench_firdemod.py 2>/dev/null | tail -1 | cut -c1-160)"
  done
done | tee $OUT/ab_fused.txt
