#!/bin/bash
# session r06j: every 15th launch of the pipelined cadences carries a bubble of about one kernel (sessions r06h, r06i: exactly 20 of 299
# gaps, whatever the look-ahead) -- is it the runtime's kernel-argument pool wrapping (HSA_KERNARG_POOL_SIZE)?  The same trace and the
# plain timing with the pool at 16 MB, and with the arguments in host memory (HIP_FORCE_DEV_KERNARG=0).
OUT=gpurun_out/r06j; mkdir -p $OUT; export TMPDIR=/tmp
for v in default pool16m hostkernarg; do
  unset HSA_KERNARG_POOL_SIZE HIP_FORCE_DEV_KERNARG
  [ $v = pool16m ] && export HSA_KERNARG_POOL_SIZE=16777216
  [ $v = hostkernarg ] && export HIP_FORCE_DEV_KERNARG=0
  rm -rf $OUT/pt
  timeout 300 rocprofv3 --kernel-trace -d $OUT/pt -o pt -f csv -- python3 tools/pipelined_trace.py > $OUT/trace_$v.json 2> $OUT/pt.err
  python3 tools/pipelined_gaps.py $OUT/pt > $OUT/gaps_$v.json
  rm -rf $OUT/pt
  python3 - $v $OUT <<'PY'
import json, sys
d = json.load(open("%s/gaps_%s.json" % (sys.argv[2], sys.argv[1])))
print(sys.argv[1], {k: (v["gaps_over_20us"], v["their_sum_us"], v["span_us_per_launch"]) for k, v in d.items()})
PY
done
