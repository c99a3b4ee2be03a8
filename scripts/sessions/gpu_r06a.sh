#!/bin/bash
# session r06a: (1) the whole GPU suite on the round-6 tree (odd-factor set-up by lane parity, FIR outputs in lane order, the streaming
# kernel's wave-per-tile instantiation); (2) LDS read patterns of the sparse operand reads (tools/ldsbench.py); (3) same-process A/B
# against the round-5 library: odd factors, FIR stores; (4) the streaming kernel: one wave per tile against four, its timeline, its
# wait-class counters.
OUT=gpurun_out/r06a; mkdir -p $OUT; export TMPDIR=/tmp
R5=rtl-sdr-rs_amd/libfmd_hip_r05.so; X=rtl-sdr-rs_amd/libfmd_hip_exp.so
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
timeout 300 python tools/ldsbench.py --json $OUT/ldsbench.jsonl > $OUT/ldsbench.txt 2>&1; tail -3 $OUT/ldsbench.txt | cut -c1-200
timeout 600 python tools/ab_libs.py --rounds 4 --cfg 5,250000,44100 --cfg 3,400000,48000 --cfg 7,166666,32000 --cfg 1,48000,48000 --cfg 9,111111,32000 --cfg ref --cfg 24 r05=$R5 new= 2>/dev/null | tee $OUT/ab_odd.txt | cut -c1-260
timeout 300 python tools/ab_libs.py --fir --rounds 4 r05=$R5 new= neworder=$X oldorder=$X@FMD_DBG=32 2>/dev/null | tee $OUT/ab_fir.txt | cut -c1-260
timeout 300 python tools/ab_libs.py --fir --rounds 4 --out-bufs 4 r05=$R5 new= neworder=$X oldorder=$X@FMD_DBG=32 2>/dev/null | tee $OUT/ab_fir_rot4.txt | cut -c1-260
timeout 300 python tools/ab_libs.py --fir --rounds 3 --fir-taps-max 127 --out-bufs 4 r05=$R5 new= 2>/dev/null | tee $OUT/ab_fir8_rot4.txt | cut -c1-260
timeout 600 python tools/ab_libs.py --rounds 4 --cfg 4,256000,48000 --cfg 4,200000,32000 --cfg 4,300000,32000 --cfg 2,500000,32000 nw4=$X nw1=$X@FMD_STREAM_NW=1 2>/dev/null | tee $OUT/ab_stream_nw.txt | cut -c1-260
timeout 300 python tools/timeline.py --cfg 4,256000,48000 --cfg 2,500000,32000 > $OUT/timeline_nw4.jsonl 2> $OUT/timeline_nw4.err; cut -c1-400 $OUT/timeline_nw4.jsonl
FMD_STREAM_NW=1 timeout 300 python tools/timeline.py --cfg 4,256000,48000 > $OUT/timeline_nw1.jsonl 2> $OUT/timeline_nw1.err; cut -c1-400 $OUT/timeline_nw1.jsonl
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE32_INSTS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_EXP_GDS SQ_WAIT_IFETCH" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_NC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum"; do
  for nw in 4 1; do
    rm -rf gpurun_out/pc
    FMD_LIB=$PWD/$X FMD_STREAM_NW=$nw timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pc -o pmc -f csv --kernel-include-regex fmd_demod -- python3 tools/bench_configs.py "D=4" > gpurun_out/pc.out 2> gpurun_out/pc.err || tail -3 gpurun_out/pc.err
    python3 - "D=4 nw=$nw" >> $OUT/pmc_stream.jsonl <<'PY'
import csv, collections, json, sys, glob
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pc/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
dur = []
for f in glob.glob('gpurun_out/pc/*kernel_trace.csv'):
    dur += [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "fmd_demod" in r["Kernel_Name"]]
w = sum(acc['SQ_WAVES']) / len(acc['SQ_WAVES']) if acc.get('SQ_WAVES') else None
print(json.dumps({"config": sys.argv[1], "waves_per_launch": w, "kernel_ns_under_pmc": round(sum(dur[-100:]) / max(1, len(dur[-100:])), 1) if dur else None,
                  "per_launch": {k: round(sum(v) / len(v), 1) for k, v in acc.items()}}))
PY
  done
done
cut -c1-600 $OUT/pmc_stream.jsonl
