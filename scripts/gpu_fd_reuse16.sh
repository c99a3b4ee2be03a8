#!/bin/bash
# Fused FIR kernel A/B on one box, interleaved (-DFMD_EXPERIMENT library):
#   decimate 8 : odd column pitch of the fragment-reuse mapping (default) against the even one (FMD_DBG bit 8)
#   decimate 16: plain mapping (default) against fragment reuse with two k-steps between the groups (FMD_FD_REUSE16)
export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
mkdir -p gpurun_out
for round in 1 2 3; do
  BENCH_FD_DECIM=8 python tools/bench_firdemod.py 2>/dev/null | tee -a gpurun_out/fd_reuse16.jsonl
  BENCH_FD_DECIM=8 FMD_DBG=256 python tools/bench_firdemod.py 2>/dev/null | tee -a gpurun_out/fd_reuse16.jsonl
  BENCH_FD_DECIM=16 python tools/bench_firdemod.py 2>/dev/null | tee -a gpurun_out/fd_reuse16.jsonl
  BENCH_FD_DECIM=16 FMD_FD_REUSE16=1 python tools/bench_firdemod.py 2>/dev/null | tee -a gpurun_out/fd_reuse16.jsonl
done
