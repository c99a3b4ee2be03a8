#!/bin/bash
# One GPU-box session that produces everything profiles/ needs for a round: tests + bench + kernel trace
# (gpu_check.sh), PMC passes (gpu_pmc.sh), the configuration table, the FIR lines.  Usage: scripts/gpu_round.sh <tag>
TAG=${1:-r02}
bash scripts/gpu_check.sh $TAG; RC=$?
bash scripts/gpu_pmc.sh ${TAG}_pmc > gpurun_out/${TAG}_pmc.log 2>&1
python3 tools/bench_configs.py > gpurun_out/${TAG}_configs.jsonl 2> gpurun_out/${TAG}_configs.err
python3 tools/bench_firdemod.py > gpurun_out/${TAG}_firdemod.json 2>/dev/null
tail -12 gpurun_out/${TAG}_configs.jsonl
exit $RC
