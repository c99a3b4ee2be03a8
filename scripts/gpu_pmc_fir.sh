#!/bin/bash
# PMC passes for the FIR kernel (config 4): one rocprofv3 run per counter set, --kernel-trace only.
TAG=${1:-pmc_fir}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export FIR_SETTLE=20
run() { # name counters...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$n -o pmc -f csv --kernel-include-regex "fmd_fir" -- python3 tools/bench_fir.py > $OUT/$n.json 2> $OUT/$n.err || tail -5 $OUT/$n.err
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
run sq3 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
python3 - $OUT <<'PY'
import csv, sys, glob, collections, json
out = {}
for n in ("sq1", "sq2", "sq3", "tcc1", "tcc2"):
    for f in glob.glob("%s/%s/*counter_collection.csv" % (sys.argv[1], n)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    for f in glob.glob("%s/%s/*kernel_trace.csv" % (sys.argv[1], n)):
        d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "fir_mfma" in r["Kernel_Name"]]
        if d:
            out.setdefault("kernel_ns_under_pmc", {})[n] = sum(d[-100:]) / len(d[-100:])
            out["kernel_name"] = [r["Kernel_Name"] for r in csv.DictReader(open(f)) if "fir_mfma" in r["Kernel_Name"]][0]
print(json.dumps(out))
open("%s/summary.json" % sys.argv[1], "w").write(json.dumps(out, indent=1))
PY
