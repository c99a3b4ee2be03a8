#!/bin/bash
# session r05al: smoke() + a default bench run on the final tree (the driver's round-end order), bench line kept
OUT=gpurun_out/r05al; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
python bench.py 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-400
