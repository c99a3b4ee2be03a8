#!/bin/bash
# session r04e: full suite; register-form kernel NG sweep with the tighter tiling; named-cvt-inside-the-store A/B; bench line
OUT=gpurun_out/r04e; mkdir -p $OUT; export TMPDIR=/tmp
echo "== full suite"
timeout 2000 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 | tee $OUT/pytest_gpu.log
echo "== firdemod NG sweep (experiment library)"
export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
for r in 1 2; do
  FMD_FD_REG=0 python tools/bench_firdemod.py 2>/dev/null | cut -c1-330
  FMD_FD_REG=5 python tools/bench_firdemod.py 2>/dev/null | cut -c1-330
  FMD_FD_REG=6 FMD_FD_LDS=23400 python tools/bench_firdemod.py 2>/dev/null | cut -c1-330
  FMD_FD_REG=7 FMD_FD_LDS=27300 python tools/bench_firdemod.py 2>/dev/null | cut -c1-330
  FMD_FD_REG=8 FMD_FD_LDS=32700 python tools/bench_firdemod.py 2>/dev/null | cut -c1-330
done | tee $OUT/fd_ab.jsonl
unset FMD_LIB
echo "== cvt A/B: builtin cast vs named instruction inside the predicated stores"
for i in 1 2 3; do bash scripts/gpu_ablibs.sh "--cfg 4,256000,48000 --cfg ref --cfg 24 --cfg 7,166666,32000" cvtb=libfmd_hip_cvtb.so new=libfmd_hip.so 2>&1 | grep -v amdgpu.ids | cut -c1-110; done | tee $OUT/ab_cvt.txt
echo "== bench"
timeout 900 python bench.py 2>$OUT/bench.err > $OUT/bench.json; cut -c1-200 $OUT/bench.json; tail -3 $OUT/bench.err
