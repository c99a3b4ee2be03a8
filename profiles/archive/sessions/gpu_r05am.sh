#!/bin/bash
# session r05am: 60 s of sustained headline launches on the final library (drift at the power cap), then the full GPU suite once more
OUT=gpurun_out/r05am; mkdir -p $OUT; export TMPDIR=/tmp
python bench.py --no-cpu --no-extra --min-timed-s 60 2>$OUT/sustained.err | tee $OUT/sustained.json | cut -c1-300
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
