#!/bin/bash
# session r04af: downsample 1 in the adjacent-sample form (one dword per lane, f32 components): parity + A/B
OUT=gpurun_out/r04af; mkdir -p $OUT; export TMPDIR=/tmp
FMD_FUZZ_CASES=150 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_variants.py tests/test_e2e_digests.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
python3 - <<'PY' 2>&1 | tail -3 | tee $OUT/d1_extra.log
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import rtl_sdr_rs_amd as fmd, oracle_lib as oracle
from test_gpu_parity import check_stream
rng = np.random.default_rng(7)
n_ok = 0
for (fast, slow) in [(48000, 48000), (250000, 48000), (1024000, 32000), (96000, 44100)]:
    for nch in (1, 9):
        blocks = [rng.integers(0, 256, (nch, 8 * int(rng.integers(3, 5000))), dtype=np.uint8) for _ in range(5)]
        blocks.append(np.where(rng.integers(0, 2, (nch, 8 * 777)) > 0, 255, 0).astype(np.uint8))
        blocks.append(np.full((nch, 8 * 300), 127, np.uint8))
        blocks.append(fmd.synth.synth_iq(nch, fmd.DEFAULT_BUF_LENGTH, amplitude=120))
        check_stream(fmd, oracle, 1, fast, slow, blocks, n_channels=nch); n_ok += 1
print("downsample-1 streams bit-exact:", n_ok)
PY
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg 1,48000,48000 --cfg 1,250000,48000 --cfg 1,1024000,32000" base=libfmd_hip_base.so new=libfmd_hip.so 2>&1 | grep '^{"cfg"'; done > $OUT/ab_d1.txt
python3 tools/ab_summary.py $OUT/ab_d1.txt
