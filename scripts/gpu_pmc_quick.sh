#!/bin/bash
# One PMC pass (instruction mix) for whichever kernel the environment selects.
OUT=gpurun_out/${1:-pmcq}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM -d $OUT/p -o pmc -f csv --kernel-include-regex "fmd_demod" -- python3 bench.py --steps 6 --warmup 2 --no-cpu > $OUT/b.json 2> $OUT/err.txt
python3 - $OUT/p/pmc_counter_collection.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list); name=""
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Counter_Name']].append(float(r['Counter_Value'])); name=r['Kernel_Name']
print(name[:60])
w = sum(acc['SQ_WAVES'])/len(acc['SQ_WAVES'])
for k, v in sorted(acc.items()):
    m = sum(v)/len(v); print("%-20s %.4g  per-wave %.1f" % (k, m, m/w))
PY
