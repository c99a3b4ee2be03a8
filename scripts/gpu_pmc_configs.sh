# Per-configuration PMC (three passes each): instruction counts, issue / wait cycles and LDS bank-conflict cycles per wave
# for tools/bench_configs.py lines.  Usage: scripts/gpu_pmc_configs.sh <tag> [config-substring ...]
# One JSON line per (configuration, pass) goes to gpurun_out/<tag>_pmc_configs.jsonl.
export TMPDIR=/tmp
TAG=${1:-pmc}; shift
OUT=gpurun_out/${TAG}_pmc_configs.jsonl
: > $OUT
if [ $# -eq 0 ]; then set -- "D=5" "cfg-ref" "D=4" "cfg-2.4"; fi
for cfg in "$@"; do
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
             "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
             "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC"; do
    rm -rf gpurun_out/pc
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pc -o pmc -f csv --kernel-include-regex fmd_demod -- python3 tools/bench_configs.py "$cfg" > /dev/null 2> gpurun_out/pc.err || tail -3 gpurun_out/pc.err
    python3 - "$cfg" >> $OUT <<'PY'
import csv, collections, json, sys, glob
acc = collections.defaultdict(list)
kern = set()
for f in glob.glob('gpurun_out/pc/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
        kern.add(r.get('Kernel_Name', '')[:60])
if acc.get('SQ_WAVES'):
    w = sum(acc['SQ_WAVES']) / len(acc['SQ_WAVES'])
    print(json.dumps({"config": sys.argv[1], "kernel": sorted(kern), "waves_per_launch": round(w),
                      "per_wave": {k: round(sum(v) / len(v) / w, 2) for k, v in acc.items() if k != 'SQ_WAVES'}}))
else:
    print(json.dumps({"config": sys.argv[1], "error": "no counters collected"}))
PY
  done
done
cat $OUT
