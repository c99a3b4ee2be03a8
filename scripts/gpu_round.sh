#!/bin/bash
# ONE GPU-box session that produces everything profiles/ needs for a round: tests + bench + kernel trace
# (gpu_check.sh), PMC passes of the headline (gpu_pmc.sh), the configuration table, per-configuration PMC (incl. the two lowest rows),
# the stand-alone and the fused FIR kernels' PMC, per-region PMC of the demodulation and the fused FIR kernels.
# Usage: scripts/gpu_round.sh <tag>;  then, here: scripts/summarize_profiles.py <tag> r06; scripts/summarize_bounds.py <tag> r06
TAG=${1:-r06}
mkdir -p gpurun_out
# the hash of every kernel family's sources as they are on THIS box (bench.family_hashes): the summaries carry them
python3 -c "import bench, json; print(json.dumps(bench.family_hashes()))" > gpurun_out/${TAG}_family_sha16.json
bash scripts/gpu_check.sh $TAG; RC=$?
bash scripts/gpu_pmc.sh ${TAG}_pmc > gpurun_out/${TAG}_pmc.log 2>&1
python3 tools/bench_configs.py > gpurun_out/${TAG}_configs.jsonl 2> gpurun_out/${TAG}_configs.err
bash scripts/gpu_pmc_configs.sh ${TAG} "cfg-ref" "cfg-2.4" "D=1 48k" "D=2 500k" "D=3 (odd)" "D=4" "D=5" "D=7" "D=8" "D=12" "D=16" "D=64" > gpurun_out/${TAG}_pmc_configs.log 2>&1
python3 tools/bench_fir.py > gpurun_out/${TAG}_fir.json 2>/dev/null
bash scripts/gpu_pmc_fir.sh ${TAG}_pmc_fir > gpurun_out/${TAG}_pmc_fir.log 2>&1
python3 tools/bench_firdemod.py > gpurun_out/${TAG}_firdemod.json 2>/dev/null
bash scripts/gpu_pmc_firdemod.sh ${TAG}_pmc_fd > gpurun_out/${TAG}_pmc_fd.log 2>&1
bash scripts/gpu_pmc_regions.sh ${TAG} > gpurun_out/${TAG}_pmc_regions.log 2>&1
bash scripts/gpu_pmc_alone.sh ${TAG} > gpurun_out/${TAG}_pmc_alone.log 2>&1
bash scripts/gpu_pmc_fd_regions.sh gpurun_out/${TAG}_pmc_fd_regions.jsonl > gpurun_out/${TAG}_pmc_fd_regions.log 2>&1
tail -16 gpurun_out/${TAG}_configs.jsonl | cut -c1-200
exit $RC
