#!/bin/bash
# session r05g: (1) staging skeletons, round-4 experiment build against this one, same process (where do the HBM-bound rows lose 0.5 - 1 %?);
# (2) the packed paired discriminator in the fused FIR kernel; (3) counters for downsample 1 and 2
OUT=gpurun_out/r05g; mkdir -p $OUT; export TMPDIR=/tmp
FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_pk.so timeout 900 python -m pytest tests/test_firdemod.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest_pk_fd.log
python tools/ab_libs.py --firdemod --rounds 5 s2= pk=rtl-sdr-rs_amd/libfmd_hip_pk.so r04=rtl-sdr-rs_amd/libfmd_hip_r04.so 2>/dev/null | tee $OUT/ab_fd_pk.jsonl | cut -c1-200
FMD_DBG=8 python tools/ab_libs.py --rounds 5 --cfg 24 --cfg ref --cfg 64,37500,8000 r04x=rtl-sdr-rs_amd/libfmd_hip_exp_r04.so s2x=rtl-sdr-rs_amd/libfmd_hip_exp.so 2>/dev/null | sed 's/^/DBG=8 /' | tee $OUT/ab_skeleton.txt | cut -c1-200
FMD_DBG=0 python tools/ab_libs.py --rounds 4 --cfg 24 --cfg 64,37500,8000 r04x=rtl-sdr-rs_amd/libfmd_hip_exp_r04.so s2x=rtl-sdr-rs_amd/libfmd_hip_exp.so 2>/dev/null | sed 's/^/DBG=0 /' | tee -a $OUT/ab_skeleton.txt | cut -c1-200
bash scripts/gpu_pmc_configs.sh r05g "D=1 48k" "D=2 500k" "cfg-ref" > $OUT/pmc_configs.log 2>&1; cut -c1-500 gpurun_out/r05g_pmc_configs.jsonl
