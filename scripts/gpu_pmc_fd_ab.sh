# instruction counts of the fused FIR kernel under environment variants: scripts/gpu_pmc_fd_ab.sh "VAR=1" "" ...
export TMPDIR=/tmp
for v in "$@"; do
rm -rf gpurun_out/pq; env $v rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU -d gpurun_out/pq -o pmc -f csv --kernel-include-regex fmd_firdemod -- python3 tools/bench_firdemod.py > /dev/null 2>&1
python3 -c "
import csv,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/pq/pmc_counter_collection.csv')): acc[r['Counter_Name']].append(float(r['Counter_Value']))
w=sum(acc['SQ_WAVES'])/len(acc['SQ_WAVES'])
print('[$v]', {k:round(sum(v)/len(v)/w,1) for k,v in acc.items() if k!='SQ_WAVES'})"
done
