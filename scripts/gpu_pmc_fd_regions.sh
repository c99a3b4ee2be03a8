#!/bin/bash
# Dynamic instruction mix per REGION of the fused FIR -> discriminator -> resampler kernel (register form, BASELINE configs[3]): PMC
# passes of the -DFMD_EXPERIMENT library with one ablation bit at a time (FMD_DBG bit 23 = staging skeleton, 16 = no operand reads + no
# matrix instructions, 17 = no matrix instructions, 18 = no digit combine / shift / conversion, 19 = no predecessor moves, 20 = no
# discriminators, 21 = no group sums, 22 = no group table); region = full - ablated.  One JSON line per ablation.
# Usage: scripts/gpu_pmc_fd_regions.sh <outfile>
export TMPDIR=/tmp
OUT=$1; : > $OUT
export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
for bit in none 23 16 17 18 19 20 21 22; do
  if [ $bit = none ]; then export FMD_DBG=0; else export FMD_DBG=$((1 << bit)); fi
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_MFMA" \
             "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"; do
    rm -rf gpurun_out/pf
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pf -o pmc -f csv --kernel-include-regex fmd_firdemod -- python3 tools/bench_firdemod.py > gpurun_out/pf.out 2> gpurun_out/pf.err || tail -3 gpurun_out/pf.err
    python3 - "$bit" >> $OUT <<'PY'
import csv, collections, json, sys, glob
acc = collections.defaultdict(list); kern = set()
for f in glob.glob('gpurun_out/pf/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Counter_Name']].append(float(r['Counter_Value'])); kern.add(r.get('Kernel_Name', '')[:90])
ms = None
try:
    ms = json.loads(open('gpurun_out/pf.out').read().splitlines()[-1]).get("ms")
except Exception:
    pass
if acc.get('SQ_WAVES'):
    w = sum(acc['SQ_WAVES']) / len(acc['SQ_WAVES'])
    print(json.dumps({"ablated_bit": sys.argv[1], "kernel": sorted(kern), "waves_per_launch": round(w), "ms_per_call_profiled": ms,
                      "per_wave": {k: round(sum(v) / len(v) / w, 2) for k, v in acc.items() if k != 'SQ_WAVES'}}))
else:
    print(json.dumps({"ablated_bit": sys.argv[1], "error": "no counters collected"}))
PY
  done
  # un-profiled time of the ablated kernel
  python3 tools/bench_firdemod.py 2>/dev/null | tail -1 | sed "s/^/{\"ablated_bit\": \"$bit\", \"timing\": /; s/$/}/" >> $OUT
done
unset FMD_LIB FMD_DBG
