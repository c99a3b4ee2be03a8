#!/bin/bash
# Per-configuration PMC (three passes each): instruction counts, issue / wait cycles and LDS bank-conflict cycles per wave
# for tools/bench_configs.py lines.  Usage: scripts/gpu_pmc_configs.sh <tag> [config-substring ...]
# One JSON line per (configuration, pass) goes to gpurun_out/<tag>_pmc_configs.jsonl; every line carries the configuration's
# (downsample, rate_out, rate_resample), the kernel the library reports, the kernel's average duration under the counters (kernel
# trace of the same run) and the hash of the kernel sources AS MEASURED (scripts/summarize_bounds.py ties the bounds to it).
export TMPDIR=/tmp
TAG=${1:-pmc}; shift
OUT=gpurun_out/${TAG}_pmc_configs.jsonl
: > $OUT
SHA=$(python3 -c "import bench; print(bench.kernel_source_hash())")
if [ $# -eq 0 ]; then set -- "D=5" "cfg-ref" "D=4" "cfg-2.4"; fi
for cfg in "$@"; do
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
             "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
             "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC"; do
    rm -rf gpurun_out/pc
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pc -o pmc -f csv --kernel-include-regex fmd_demod -- python3 tools/bench_configs.py "$cfg" > gpurun_out/pc.out 2> gpurun_out/pc.err || tail -3 gpurun_out/pc.err
    python3 - "$cfg" "$SHA" >> $OUT <<'PY'
import csv, collections, json, sys, glob
acc = collections.defaultdict(list)
kern = set()
for f in glob.glob('gpurun_out/pc/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
        kern.add(r.get('Kernel_Name', '')[:60])
dur = []
for f in glob.glob('gpurun_out/pc/*kernel_trace.csv'):
    dur += [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "fmd_demod" in r["Kernel_Name"]]
line = {}
try:
    line = json.loads([l for l in open('gpurun_out/pc.out').read().splitlines() if l.startswith('{"config"')][-1])
except Exception:
    pass
if acc.get('SQ_WAVES'):
    w = sum(acc['SQ_WAVES']) / len(acc['SQ_WAVES'])
    print(json.dumps({"config": sys.argv[1], "cfg": line.get("cfg"), "kernel_reported": line.get("kernel"), "kernel": sorted(kern), "waves_per_launch": round(w),
                      "kernel_ns_under_pmc": round(sum(dur[-100:]) / max(1, len(dur[-100:])), 1) if dur else None, "kernel_source_sha16": sys.argv[2],
                      "per_wave": {k: round(sum(v) / len(v) / w, 2) for k, v in acc.items() if k != 'SQ_WAVES'}}))
else:
    print(json.dumps({"config": sys.argv[1], "error": "no counters collected"}))
PY
  done
done
cat $OUT
