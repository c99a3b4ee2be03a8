#!/bin/bash
# Cache policy of the staging loads (aux of global_load_lds: 1 sc0, 2 nt, 16 sc1) against time AND power: libraries built
# with  make -C rtl-sdr-rs_amd/csrc ab ABFLAGS="-DFMD_EXPERIMENT -DFMD_DMA_AUX=<n>" ABNAME=aux<n>;  aux 2 is the shipped policy.
mkdir -p gpurun_out
for n in exp aux0 aux3 aux17 aux19 exp; do
  FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_$n.so python tools/clock_probe.py --cfg ${1:-24} --seconds 2 stage_only:8 full:0 2>/dev/null | grep -v idle | sed "s/^{/{\"lib\": \"$n\", /" | tee -a gpurun_out/aux_power.jsonl
done
