#!/usr/bin/env python3
"""Shader clock and socket power WHILE a kernel variant runs (tuning tool, not part of the product).

Question it answers: is a configuration whose memory side and compute side are both busy slower than
max(loads only, compute only) because the chip lowers its clock under the combined load?  For every variant (ablation
bits of the -DFMD_EXPERIMENT library, FMD_DBG) the kernel is launched back to back for `--seconds` while a thread samples
the device's current shader clock and average socket power (bench.GpuSensors: amdgpu sysfs, matched by PCI address);
the launch time comes from HIP events over the same interval.  One JSON line per variant.  Usage:
    FMD_LIB=rtl-sdr-rs_amd/libfmd_hip_exp.so tools/clock_probe.py --cfg ref full:0 stage_only:8 no_loads:16 no_rounds:64
    FMD_LIB=... tools/clock_probe.py --firdemod full:0 no_mfma:1 no_disc:2 neither:3     (BASELINE config 4, fused kernel)
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import rtl_sdr_rs_amd as fmd

NAMED = {"ref": (6, 170000, 32000), "24": (10, 240000, 32000)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="ref")
    ap.add_argument("--firdemod", action="store_true", help="the fused FIR kernel at BASELINE config 4 instead of the boxcar kernel")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--channels", type=int, default=4096)
    ap.add_argument("variants", nargs="+", help="name:FMD_DBG value")
    a = ap.parse_args()
    sens = bench.GpuSensors(torch, 0)
    stream = torch.cuda.current_stream().cuda_stream
    if a.firdemod:
        nch, N = 256, 2 << 20
    else:
        nch, N = a.channels, fmd.DEFAULT_BUF_LENGTH
        D, fast, slow = NAMED[a.cfg] if a.cfg in NAMED else tuple(int(x) for x in a.cfg.split(","))
    bufs = []
    for b in range(3):
        t = torch.empty((nch, N), dtype=torch.uint8, device="cuda")
        fmd.synth.fill_device(t.data_ptr(), nch, N, sample_offset=b * (N // 2), stream=stream)
        bufs.append(t)
    torch.cuda.synchronize()
    time.sleep(0.5)
    s, p = sens.sample_while(lambda: time.sleep(0.05), 1.0)
    print(json.dumps({"variant": "idle", "sysfs": sens.dir, "sclk_mhz": bench.sensor_stats(s), "power_w": bench.sensor_stats(p),
                      "power_cap_w": sens.power_cap_w(), "sclk_max_level_mhz": sens.sclk_max_mhz()}), flush=True)
    for v in a.variants:
        name, _, dbg = v.partition(":")
        os.environ["FMD_DBG"] = dbg or "0"
        if a.firdemod:
            taps = np.random.default_rng(1).integers(-2047, 2048, 127).astype(np.int16)
            bank = fmd.FirDemodBank(taps, 8, 2500000, 48000, nch)
            cap = bank.out_cap(N)
            out = torch.zeros((nch, cap), dtype=torch.int16, device="cuda")
            call = lambda i: bank.demodulate_device(bufs[i % 3].data_ptr(), N, out.data_ptr(), cap, stream)
            what = "fmd_firdemod_kernel, config 4"
        else:
            cfg = fmd.DemodConfig(fast, fast, slow, D, max(1, (1 << 15) // (128 * D)))
            bank = fmd.DemodBank(cfg, nch)
            cap = bank.out_cap(N)
            out = torch.zeros((nch, cap), dtype=torch.int16, device="cuda")
            call = lambda i: bank.demodulate_device(bufs[i % 3].data_ptr(), N, out.data_ptr(), cap, None, stream)
            what = [D, fast, slow]
        for i in range(300):
            call(i)
        torch.cuda.synchronize()
        ms = []

        def work():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(500):
                call(i)
            e1.record(); torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / 500)

        sclk, power = sens.sample_while(work, a.seconds)
        n6 = max(1, len(ms) // 6)                            # drift over the interval: medians of six consecutive slices
        slices = [round(sorted(ms[i:i + n6])[len(ms[i:i + n6]) // 2], 4) for i in range(0, n6 * 6, n6) if ms[i:i + n6]]
        sc6 = [sorted(sclk[i:i + max(1, len(sclk) // 6)])[len(sclk[i:i + max(1, len(sclk) // 6)]) // 2] for i in range(0, max(1, len(sclk) // 6) * 6, max(1, len(sclk) // 6)) if sclk[i:i + max(1, len(sclk) // 6)]]
        ms.sort()
        print(json.dumps({"variant": name, "ms_per_call_over_time": slices, "sclk_mhz_over_time": sc6, "FMD_DBG": dbg, "kernel": what, "launches": 500 * len(ms), "ms_per_call_median": round(ms[len(ms) // 2], 4),
                          "sclk_mhz": bench.sensor_stats(sclk), "power_w": bench.sensor_stats(power)}), flush=True)
        del bank, out
        time.sleep(1.0)


if __name__ == "__main__":
    main()
