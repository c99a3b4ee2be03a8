#!/bin/bash
# session r04o: tile-size sweep around the shipped tiling on the final kernels (experiment library, FMD_KT)
OUT=gpurun_out/r04o; mkdir -p $OUT; export TMPDIR=/tmp
export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab.py --rounds 3 --cfg 24 k118: k106:FMD_KT=106 k110:FMD_KT=110 k114:FMD_KT=114 k122:FMD_KT=122 k126:FMD_KT=126 2>/dev/null | grep '^{"cfg"' > $OUT/kt24.txt
python tools/ab.py --rounds 3 --cfg ref k256: k192:FMD_KT=192 k208:FMD_KT=208 k224:FMD_KT=224 k240:FMD_KT=240 k248:FMD_KT=248 2>/dev/null | grep '^{"cfg"' > $OUT/ktref.txt
python tools/ab.py --rounds 3 --cfg 7,166666,32000 kdef: k192:FMD_KT=192 k224:FMD_KT=224 k256:FMD_KT=256 2>/dev/null | grep '^{"cfg"' > $OUT/kt7.txt
python tools/ab.py --rounds 3 --cfg 5,250000,44100 kdef: k192:FMD_KT=192 k224:FMD_KT=224 k288:FMD_KT=288 2>/dev/null | grep '^{"cfg"' > $OUT/kt5.txt
unset FMD_LIB
python3 tools/ab_summary.py $OUT/kt24.txt; python3 tools/ab_summary.py $OUT/ktref.txt; python3 tools/ab_summary.py $OUT/kt7.txt; python3 tools/ab_summary.py $OUT/kt5.txt
