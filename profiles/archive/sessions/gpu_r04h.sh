#!/bin/bash
# session r04h: boxcar tiles beyond the 8-per-CU budget (experiment library, FMD_KT); full suite on the final sources
OUT=gpurun_out/r04h; mkdir -p $OUT; export TMPDIR=/tmp
echo "== boxcar tile size sweep"
export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab.py --rounds 2 --cfg ref k256: k320:FMD_KT=320 k380:FMD_KT=380 k508:FMD_KT=508 2>&1 | grep -v amdgpu.ids | cut -c1-160 | tee $OUT/kt_ref.txt
python tools/ab.py --rounds 2 --cfg 24 k118: k150:FMD_KT=150 k178:FMD_KT=178 k236:FMD_KT=236 2>&1 | grep -v amdgpu.ids | cut -c1-160 | tee $OUT/kt_24.txt
unset FMD_LIB
echo "== full suite"
timeout 2000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest_gpu.log
