#!/bin/bash
# PMC passes for the demod kernel (each pass its own rocprofv3 run; --kernel-trace only).
TAG=${1:-pmc}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --settle 20 --no-cpu --no-extra $@"
# the hash of the kernel sources AS MEASURED (summarize_profiles.py ties the summary to it; bench.py quotes the traffic only
# when its own sources hash the same)
python3 -c "import bench; print(bench.kernel_source_hash())" > $OUT/kernel_source_sha16.txt
run() { # name counters...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$n -o pmc -f csv --kernel-include-regex "fmd_demod" -- python3 bench.py $ARGS > $OUT/$n.json 2> $OUT/$n.err || tail -5 $OUT/$n.err
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
for n in sq1 sq2 tcc1 tcc2 grbm; do f=$(ls $OUT/$n/*counter_collection.csv 2>/dev/null | head -1); echo "== $n $f"; [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print("%-24s n=%d mean=%.6g" % (k, len(v), sum(v) / len(v)))
PY
done
