# Per-configuration PMC (two passes each): instruction counts and issue / wait cycles per wave for tools/bench_configs.py lines.
export TMPDIR=/tmp
for cfg in ${@:-"D=5" "cfg-ref" "D=7" "cfg-2.4" "D=16"}; do
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
    rm -rf gpurun_out/pc; rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pc -o pmc -f csv --kernel-include-regex fmd_demod -- python3 tools/bench_configs.py "$cfg" > /dev/null 2>&1
    python3 -c "
import csv,collections,sys
acc=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/pc/pmc_counter_collection.csv')): acc[r['Counter_Name']].append(float(r['Counter_Value']))
w=sum(acc['SQ_WAVES'])/len(acc['SQ_WAVES'])
print('$cfg', round(w), {k:round(sum(v)/len(v)/w,1) for k,v in acc.items() if k!='SQ_WAVES'})"
  done
done
