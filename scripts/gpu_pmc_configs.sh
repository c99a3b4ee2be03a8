export TMPDIR=/tmp
for cfg in "D=64" "D=16" "D=8" "cfg-2.4" "D=4"; do
  rm -rf gpurun_out/pc; rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d gpurun_out/pc -o pmc -f csv --kernel-include-regex fmd_demod -- python3 tools/bench_configs.py "$cfg" > /dev/null 2>&1
  python3 -c "
import csv,collections,sys
acc=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/pc/pmc_counter_collection.csv')): acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('$cfg', {k:round(sum(v)/len(v)) for k,v in acc.items()})"
done
