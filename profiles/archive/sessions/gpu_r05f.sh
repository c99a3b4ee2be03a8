#!/bin/bash
# session r05f: (1) the paired packed-f32 discriminator (-DFMD_PK_DISC build) against the shipped form, same process; (2) where the
# kernel arguments live (HIP_FORCE_DEV_KERNARG 0 / 1): the 64-byte rows made the HBM-bound rows 0.6 - 1.3 % slower in r05e
OUT=gpurun_out/r05f; mkdir -p $OUT; export TMPDIR=/tmp
FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_pk.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest_pk.log
python tools/ab_libs.py --rounds 5 --cfg 24 --cfg ref --cfg 4,256000,48000 --cfg 8,250000,44100 --cfg 2,500000,32000 --cfg 12,192000,32000 s2= pk=rtl-sdr-rs_amd/libfmd_hip_pk.so 2>/dev/null | tee $OUT/ab_pk.jsonl | cut -c1-230
for ka in 0 1; do
  HIP_FORCE_DEV_KERNARG=$ka python tools/ab_libs.py --rounds 4 --cfg 24 --cfg ref --cfg 64,37500,8000 r04=rtl-sdr-rs_amd/libfmd_hip_r04.so s2= 2>/dev/null | sed "s/^/KA=$ka /" | tee -a $OUT/ab_kernarg.txt | cut -c1-230
done
python tools/ab_libs.py --rounds 4 --cfg 24 --cfg ref --cfg 64,37500,8000 r04=rtl-sdr-rs_amd/libfmd_hip_r04.so s2= 2>/dev/null | sed "s/^/KA=default /" | tee -a $OUT/ab_kernarg.txt | cut -c1-230
# (3) per-region counters of the fused FIR kernel
bash scripts/gpu_pmc_fd_regions.sh $OUT/fd_regions.jsonl > $OUT/fd_regions.log 2>&1; cut -c1-420 $OUT/fd_regions.jsonl
