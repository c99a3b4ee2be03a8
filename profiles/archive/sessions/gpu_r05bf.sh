#!/bin/bash
# session r05bf: the FIR kernels' share of the round's evidence again (stand-alone kernel changed: sparse two-digit form), then smoke + bench
export TMPDIR=/tmp
TAG=r05
python3 tools/bench_fir.py > gpurun_out/${TAG}_fir.json 2>/dev/null
bash scripts/gpu_pmc_fir.sh ${TAG}_pmc_fir > gpurun_out/${TAG}_pmc_fir.log 2>&1
python3 tools/bench_firdemod.py > gpurun_out/${TAG}_firdemod.json 2>/dev/null
bash scripts/gpu_pmc_firdemod.sh ${TAG}_pmc_fd > gpurun_out/${TAG}_pmc_fd.log 2>&1
cat gpurun_out/${TAG}_fir.json | cut -c1-300
mkdir -p gpurun_out/r05al
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r05al/smoke.log
python bench.py 2>gpurun_out/r05al/bench.err | tee gpurun_out/r05al/bench.json | cut -c1-300
