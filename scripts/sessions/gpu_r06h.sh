#!/bin/bash
# session r06h: (1) kernel durations and inter-kernel gaps of the three completion cadences from a kernel trace (where the pipelined
# completion point's last few percent go); (2) the evidence session (r06g stopped at a too tight bound in tests/test_bench_launch.py:
# check_pipelined read 1.067 x on that box) and the bench line that quotes its summaries.
export TMPDIR=/tmp; mkdir -p gpurun_out/r06h
rm -rf gpurun_out/r06h/pt
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/r06h/pt -o pt -f csv -- python3 tools/pipelined_trace.py > gpurun_out/r06h/pipelined_trace.json 2> gpurun_out/r06h/pt.err
python3 tools/pipelined_gaps.py gpurun_out/r06h/pt > gpurun_out/r06h/pipelined_gaps.json; cat gpurun_out/r06h/pipelined_trace.json gpurun_out/r06h/pipelined_gaps.json
rm -rf gpurun_out/r06h/pt
bash scripts/sessions/gpu_r06g.sh
