#!/bin/bash
# session r05u: smoke() + the full GPU suite + a default bench run, on the final tree (what the driver runs at round end)
OUT=gpurun_out/r05u; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.log
python bench.py 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-300
