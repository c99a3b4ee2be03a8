#!/bin/bash
# session r06b: (1) the GPU suite on the round-6 tree (pipelined completion point, permuted FIR tile image, half-swapped sparse operand
# reads); (2) same-process A/B of both FIR kernels against round 5; (3) the streaming kernel with 16 instead of 12 rounds per wave
# (second lever on fmd_demod_stream_kernel<2, 2>); (4) LDS conflict counters of both FIR kernels; (5) the default bench line;
# (6) the vector pipe's ceiling on the downsample-1 kernel (compute on resident LDS, no loads) for the bound model's calibration.
OUT=gpurun_out/r06b; mkdir -p $OUT; export TMPDIR=/tmp
R5=rtl-sdr-rs_amd/libfmd_hip_r05.so; X=rtl-sdr-rs_amd/libfmd_hip_exp.so; R16=rtl-sdr-rs_amd/libfmd_hip_r16.so
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -15 $OUT/pytest.txt
timeout 300 python tools/ab_libs.py --firdemod --rounds 4 r05=$R5 new= swap=$X noswap=$X@FMD_DBG=536870912 2>/dev/null | tee $OUT/ab_fd.txt | cut -c1-260
timeout 300 python tools/ab_libs.py --firdemod --rounds 4 --fir-taps-max 127 r05=$R5 new= 2>/dev/null | tee $OUT/ab_fd8.txt | cut -c1-260
timeout 300 python tools/ab_libs.py --fir --rounds 4 r05=$R5 new= 2>/dev/null | tee $OUT/ab_fir.txt | cut -c1-260
timeout 300 python tools/ab_libs.py --fir --rounds 4 --out-bufs 4 r05=$R5 new= 2>/dev/null | tee $OUT/ab_fir_rot4.txt | cut -c1-260
timeout 300 python tools/ab_libs.py --fir --rounds 4 --out-bufs 4 --fir-taps-max 127 r05=$R5 new= 2>/dev/null | tee $OUT/ab_fir8_rot4.txt | cut -c1-260
timeout 300 python tools/ab_libs.py --fir --rounds 4 --fir-taps-max 127 r05=$R5 new= 2>/dev/null | tee $OUT/ab_fir8.txt | cut -c1-260
timeout 600 python tools/ab_libs.py --rounds 4 --cfg 4,256000,48000 --cfg 4,200000,32000 --cfg 4,300000,32000 --cfg 2,500000,32000 --cfg 2,96000,48000 r12=$X r16=$R16 2>/dev/null | tee $OUT/ab_stream_r16.txt | cut -c1-260
for k in fir firdemod; do
  rm -rf gpurun_out/pc
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA -d gpurun_out/pc -o pmc -f csv --kernel-include-regex fmd_fir -- python3 tools/bench_$k.py > gpurun_out/pc.out 2> gpurun_out/pc.err || tail -3 gpurun_out/pc.err
  python3 - "$k" >> $OUT/pmc_lds.jsonl <<'PY'
import csv, collections, json, sys, glob
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pc/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print(json.dumps({"kernel": sys.argv[1], "per_launch": m, "lds_bank_conflict_share": m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"] if m.get("SQ_LDS_IDX_ACTIVE") else None}))
PY
done
cat $OUT/pmc_lds.jsonl | cut -c1-400
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python3 -c "
import json; d=json.load(open('$OUT/bench.json')); e=d['extra']
print(d['ms_per_step'], d['roofline']['frac'], {k:e[k].get('ms_per_step') for k in ('check_per_step','check_pipelined')}, e['cfg_ref'].get('frac'), e['config4_fir'].get('frac'), e['config4_fir'].get('one_output_buffer'), e['config4_fir_demod_fused'].get('frac'), e['config4_fir_demod_fused'].get('taps_8bit_one_digit'))
for r in e['domain']['rows']: print(r['downsample'], r['frac'], r['kernel'])
" 2>&1 | cut -c1-300
for dbg in 0 16; do
  rm -rf gpurun_out/pc
  FMD_LIB=$PWD/$X FMD_DBG=$dbg timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_LDS -d gpurun_out/pc -o pmc -f csv --kernel-include-regex fmd_demod -- python3 tools/bench_configs.py "D=1" "D=3" "cfg-ref" > gpurun_out/pc.out 2> gpurun_out/pc.err || tail -3 gpurun_out/pc.err
  python3 - "$dbg" >> $OUT/pmc_compute_alone.jsonl <<'PY'
import csv, collections, json, sys, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pc/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    print(json.dumps({"dbg": int(sys.argv[1]), "kernel": k, "per_launch": {n: sum(v) / len(v) for n, v in c.items()}}))
PY
  grep '^{"config"' gpurun_out/pc.out | cut -c1-200 >> $OUT/compute_alone_lines.txt
done
cut -c1-400 $OUT/pmc_compute_alone.jsonl
