#!/bin/bash
# session r05q: the committed bench line (made AFTER the PMC summary / bounds it quotes) + timing of the whole default run
OUT=gpurun_out/r05q; mkdir -p $OUT; export TMPDIR=/tmp
T0=$(date +%s.%N); python bench.py > $OUT/bench.json 2> $OUT/bench.err; T1=$(date +%s.%N); echo "bench.py wall seconds: $(echo "$T1 - $T0" | bc)"
python3 -c "
import json
r=json.load(open('$OUT/bench.json'))
print(r['ms_per_step'], r['roofline']['frac'], r['roofline']['traffic'], r['roofline'].get('valu_issue_frac'), r['roofline'].get('frac_of_skeleton'))
print('cfg_ref', r['extra']['cfg_ref']['ms_per_call'], r['extra']['cfg_ref']['frac'], r['extra']['cfg_ref'].get('bound'), r['extra']['cfg_ref'].get('frac_of_skeleton'))
for row in r['extra']['domain']['rows']: print(row['downsample'], row['ms_per_call'], row['frac'], row.get('bound'), row.get('valu_issue_frac'), row.get('salu_issue_frac'))
print(r['extra']['config4_fir'].get('bound'), r['extra']['config4_fir'].get('mfma_busy_frac'), r['extra']['config4_fir_demod_fused'].get('bound'), r['extra']['config4_fir_demod_fused'].get('valu_issue_frac'), r['extra']['config4_fir_demod_fused'].get('mfma_busy_frac'))
"
