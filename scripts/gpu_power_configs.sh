#!/bin/bash
# Shader clock and socket power of the SHIPPED library at every configuration of tools/bench_configs.py (tools/clock_probe.py,
# 2 s of back-to-back launches each): which configurations run at the board's power cap, and at what clock.
mkdir -p gpurun_out
: > gpurun_out/power_configs.jsonl
for cfg in 6,170000,32000 10,240000,32000 4,256000,48000 8,250000,44100 2,500000,32000 7,166666,32000 5,250000,44100 1,48000,48000 16,150000,32000 64,37500,8000 32,512000,32000 12,192000,32000 13,208000,32000 14,224000,32000; do
  python tools/clock_probe.py --cfg $cfg --seconds 2 shipped:0 2>/dev/null | grep '"shipped"' | tee -a gpurun_out/power_configs.jsonl
done
