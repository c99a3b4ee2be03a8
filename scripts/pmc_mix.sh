#!/bin/bash
# One PMC pass (instruction mix per wave) of tools/bench_configs.py lines.  Usage: scripts/pmc_mix.sh <outfile> <config-substring>...
# (the library is whatever FMD_LIB selects)
export TMPDIR=/tmp
OUT=$1; shift
for cfg in "$@"; do
  rm -rf gpurun_out/pm
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD -d gpurun_out/pm -o pmc -f csv --kernel-include-regex fmd_demod -- python3 tools/bench_configs.py "$cfg" > /dev/null 2> gpurun_out/pm.err || tail -3 gpurun_out/pm.err
  python3 - "$cfg" >> $OUT <<'PY'
import csv, collections, json, sys, glob, os
acc = collections.defaultdict(list); kern = set()
for f in glob.glob('gpurun_out/pm/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Counter_Name']].append(float(r['Counter_Value'])); kern.add(r.get('Kernel_Name', '')[:60])
if acc.get('SQ_WAVES'):
    w = sum(acc['SQ_WAVES']) / len(acc['SQ_WAVES'])
    print(json.dumps({"config": sys.argv[1], "lib": os.path.basename(os.environ.get("FMD_LIB", "libfmd_hip.so")), "kernel": sorted(kern),
                      "per_wave": {k: round(sum(v) / len(v) / w, 1) for k, v in acc.items() if k != 'SQ_WAVES'}}))
else:
    print(json.dumps({"config": sys.argv[1], "error": "no counters"}))
PY
done
