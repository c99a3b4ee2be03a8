#!/bin/bash
# session r06g: the evidence session again on the final sources (the opt-in event ordering touched fmd_api.cpp / fmd_host.h, which every
# kernel family's hash covers): scripts/gpu_round.sh r06, then -- with the summaries made ON THE BOX from this session's counters --
# the bench line that quotes them.
bash scripts/gpu_round.sh r06
python3 scripts/summarize_profiles.py r06 r06 > gpurun_out/r06_summarize.log 2>&1
python3 scripts/summarize_bounds.py r06 r06 >> gpurun_out/r06_summarize.log 2>&1
mkdir -p gpurun_out/r06g
timeout 600 python bench.py > gpurun_out/r06g/bench.json 2> gpurun_out/r06g/bench.err; tail -2 gpurun_out/r06g/bench.err
cp profiles/r06_*.json profiles/r06_*.jsonl profiles/r06_*.csv gpurun_out/r06g/ 2>/dev/null
