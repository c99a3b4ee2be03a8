#!/bin/bash
# The two sides of every LDS-tile row ALONE (round 6: a bound model that closes by construction): the -DFMD_EXPERIMENT library with the
# staging loads ablated (FMD_DBG=16: the rounds and the resampler run on whatever the LDS holds -- the compute side alone) against the
# same library unablated, all configurations in ONE process per pass, counters SQ_BUSY_CYCLES / SQ_INSTS_VALU / SQ_INSTS_SALU.
# scripts/summarize_bounds.py turns them into `compute_alone_cycles_frac` = shader clocks of the compute side alone / of the whole
# kernel (clock-independent: both in shader clocks).   Usage: scripts/gpu_pmc_alone.sh <tag>
export TMPDIR=/tmp
TAG=${1:-r06}
OUT=gpurun_out/${TAG}_pmc_alone.jsonl
: > $OUT
for dbg in 0 16; do
  rm -rf gpurun_out/pa
  FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so FMD_DBG=$dbg timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_LDS \
      -d gpurun_out/pa -o pmc -f csv --kernel-include-regex fmd_demod -- python3 tools/bench_configs.py > gpurun_out/pa.out 2> gpurun_out/pa.err || tail -3 gpurun_out/pa.err
  python3 - "$dbg" >> $OUT <<'PY'
import csv, collections, json, sys, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pa/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    name = k[5:] if k.startswith("void ") else k
    name = name.split("(FmdLaunch)")[0]
    print(json.dumps({"dbg": int(sys.argv[1]), "kernel": name, "per_launch": {n: sum(v) / len(v) for n, v in c.items()}}))
PY
done
cut -c1-200 $OUT
