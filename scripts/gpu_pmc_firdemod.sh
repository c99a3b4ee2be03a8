#!/bin/bash
# PMC passes for the fused FIR -> discriminator -> resampler kernel (each pass its own rocprofv3 run).
TAG=${1:-pmc_fd}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$n -o pmc -f csv --kernel-include-regex "fmd_firdemod" -- python3 tools/bench_firdemod.py > $OUT/$n.json 2> $OUT/$n.err || tail -5 $OUT/$n.err
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
for n in sq1 sq2 sq3 tcc1 tcc2 grbm; do f=$(ls $OUT/$n/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print("%-26s n=%d mean=%.6g" % (k, len(v), sum(v) / len(v)))
PY
done
python3 - $OUT <<'PY'
import csv, sys, glob, collections, json
out = {}
for n in ("sq1", "sq2", "sq3", "tcc1", "tcc2", "grbm"):
    for f in glob.glob("%s/%s/*counter_collection.csv" % (sys.argv[1], n)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    for f in glob.glob("%s/%s/*kernel_trace.csv" % (sys.argv[1], n)):
        d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "firdemod" in r["Kernel_Name"]]
        if d:
            out.setdefault("kernel_ns_under_pmc", {})[n] = sum(d) / len(d)
open("%s/summary.json" % sys.argv[1], "w").write(json.dumps(out, indent=1))
PY
