#!/bin/bash
# A/B of library BUILDS on one box: scripts/gpu_ablibs.sh "<ab.py args>" name1=lib1.so name2=lib2.so ...  (two alternating rounds)
ARGS=$1; shift
for r in 1 2; do
  for v in "$@"; do
    n=${v%%=*}; l=${v#*=}
    FMD_LIB=$PWD/rtl-sdr-rs_amd/$l python tools/ab.py $ARGS --rounds 1 $n: | cut -c1-120
  done
done
