#!/bin/bash
# session r06d: the round's evidence session -- scripts/gpu_round.sh r06 (GPU suite, bench line, kernel trace, PMC passes of the headline,
# of every configuration row, of both FIR kernels, per-region passes, the compute side alone) on the final kernels.
bash scripts/gpu_round.sh r06
