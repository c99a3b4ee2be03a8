#!/bin/bash
# A/B build of the library with extra compiler flags (a temporary -D switch while an experiment is open):
#   scripts/build_variant.sh <name> [flags...]   ->  rtl-sdr-rs_amd/libfmd_hip_<name>.so   (load it with FMD_LIB=...)
# Objects go to csrc/build/<name>/ (git-ignored).  The shipped sources carry no such switches once an experiment is closed.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../rtl-sdr-rs_amd/csrc"
SRCS=$(sed -n 's/^SRCS *:= *//p' Makefile)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wextra -Wno-unused-parameter -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form"
mkdir -p build/$NAME
pids=()
for s in $SRCS; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -x hip -c $s -o build/$NAME/$s.o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/$NAME/*.o -o ../libfmd_hip_$NAME.so
ls -la ../libfmd_hip_$NAME.so
