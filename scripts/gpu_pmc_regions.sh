#!/bin/bash
# Dynamic instruction mix per REGION of the demodulation kernel (VERDICT r3 #3a): PMC passes of the -DFMD_EXPERIMENT
# library with one ablation bit at a time (FMD_DBG: 8 = staging skeleton only, 64 = no boxcar + discriminator rounds,
# 128 = no resampler pass, 32 = no state epilogue); region = full - ablated.  One JSON line per (configuration, ablation)
# goes to gpurun_out/<tag>_pmc_regions.jsonl.   Usage: scripts/gpu_pmc_regions.sh <tag> [config-substring ...]
export TMPDIR=/tmp
TAG=${1:-pmc}; shift
OUT=gpurun_out/${TAG}_pmc_regions.jsonl
: > $OUT
if [ $# -eq 0 ]; then set -- "cfg-ref" "cfg-2.4"; fi
export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
for cfg in "$@"; do
  for dbg in 0 8 64 128 32; do
    export FMD_DBG=$dbg
    rm -rf gpurun_out/pr
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM \
        -d gpurun_out/pr -o pmc -f csv --kernel-include-regex fmd_demod -- python3 tools/bench_configs.py "$cfg" > gpurun_out/pr.out 2> gpurun_out/pr.err || tail -3 gpurun_out/pr.err
    python3 - "$cfg" "$dbg" >> $OUT <<'PY'
import csv, collections, json, sys, glob
acc = collections.defaultdict(list)
kern = set()
for f in glob.glob('gpurun_out/pr/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
        kern.add(r.get('Kernel_Name', '')[:70])
ms = None
try:
    ms = json.loads(open('gpurun_out/pr.out').read().splitlines()[-1]).get("ms_per_call")
except Exception:
    pass
if acc.get('SQ_WAVES'):
    w = sum(acc['SQ_WAVES']) / len(acc['SQ_WAVES'])
    print(json.dumps({"config": sys.argv[1], "FMD_DBG": int(sys.argv[2]), "kernel": sorted(kern), "waves_per_launch": round(w),
                      "ms_per_call_profiled": ms, "per_wave": {k: round(sum(v) / len(v) / w, 2) for k, v in acc.items() if k != 'SQ_WAVES'}}))
else:
    print(json.dumps({"config": sys.argv[1], "FMD_DBG": int(sys.argv[2]), "error": "no counters collected"}))
PY
  done
done
unset FMD_LIB FMD_DBG
cat $OUT
