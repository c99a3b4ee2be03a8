export TMPDIR=/tmp
for d in 0 1 2 4 7; do
  rm -rf gpurun_out/pa; FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so FMD_DBG=$d rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -d gpurun_out/pa -o pmc -f csv --kernel-include-regex fmd_firdemod -- python3 tools/bench_firdemod.py > /dev/null 2>&1
  python3 -c "
import csv,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/pa/pmc_counter_collection.csv')): acc[r['Counter_Name']].append(float(r['Counter_Value']))
w=sum(acc['SQ_WAVES'])/len(acc['SQ_WAVES'])
print('dbg $d', {k:round(sum(v)/len(v)/w,1) for k,v in acc.items()})"
done
