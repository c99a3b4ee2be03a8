#!/bin/bash
# session r04z: tile-size sweep on the kernels after the discriminator changes (experiment library, FMD_KT)
OUT=gpurun_out/r04z; mkdir -p $OUT; export TMPDIR=/tmp
export FMD_LIB=$PWD/rtl-sdr-rs_amd/libfmd_hip_exp.so
python tools/ab.py --rounds 3 --cfg 24 k118: k102:FMD_KT=102 k110:FMD_KT=110 k126:FMD_KT=126 k134:FMD_KT=134 2>/dev/null | grep '^{"cfg"' > $OUT/kt24.txt
python tools/ab.py --rounds 3 --cfg ref k256: k208:FMD_KT=208 k240:FMD_KT=240 k272:FMD_KT=272 k304:FMD_KT=304 2>/dev/null | grep '^{"cfg"' > $OUT/ktref.txt
python tools/ab.py --rounds 3 --cfg 8,250000,44100 kdef: k160:FMD_KT=160 k200:FMD_KT=200 k220:FMD_KT=220 2>/dev/null | grep '^{"cfg"' > $OUT/kt8.txt
python tools/ab.py --rounds 3 --cfg 5,250000,44100 kdef: k224:FMD_KT=224 k288:FMD_KT=288 k320:FMD_KT=320 2>/dev/null | grep '^{"cfg"' > $OUT/kt5.txt
python tools/ab.py --rounds 3 --cfg 12,192000,32000 kdef: k100:FMD_KT=100 k140:FMD_KT=140 k160:FMD_KT=160 2>/dev/null | grep '^{"cfg"' > $OUT/kt12.txt
unset FMD_LIB
for f in kt24 ktref kt8 kt5 kt12; do python3 tools/ab_summary.py $OUT/$f.txt; done
