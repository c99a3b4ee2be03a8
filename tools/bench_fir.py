#!/usr/bin/env python3
"""BASELINE configs[3] measurement (not the headline bench): 127-tap FIR + 8x decimate, 256 channels at
20 Msps -- one call = 2 MiB per channel (52.4 ms of signal), inputs resident in HBM.  Prints one JSON line.
FMD_FIR_MFMA=0 times the VALU form instead of the matrix-core form."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    import rtl_sdr_rs_amd as fmd
    nch, n, T, M = 256, 2 << 20, 127, 8
    rng = np.random.default_rng(1)
    taps = rng.integers(-2047, 2048, T).astype(np.int16)
    bank = fmd.FirBank(taps, M, nch)
    dev = torch.device("cuda", 0)
    bufs = []
    stream = torch.cuda.current_stream().cuda_stream
    for b in range(2):
        t = torch.empty((nch, n), dtype=torch.uint8, device=dev)
        fmd.synth.fill_device(t.data_ptr(), nch, n, sample_offset=b * (n // 2), stream=stream)
        bufs.append(t)
    cap = bank.out_cap(n)
    out = torch.zeros((nch, cap, 2), dtype=torch.int32, device=dev)
    for i in range(int(os.environ.get('FIR_SETTLE', '150'))):     # untimed: clock ramp after idle
        bank.filter_device(bufs[i % 2].data_ptr(), n, out.data_ptr(), cap, stream)
    torch.cuda.synchronize()
    steps = 100
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        nout = bank.filter_device(bufs[i % 2].data_ptr(), n, out.data_ptr(), cap, stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    samples = nch * (n // 2)
    alg = nch * n + nch * nout * 8
    macs = nch * nout * T * 2
    print(json.dumps({
        "workload": "BASELINE configs[3]: %d-tap FIR, decimate %d, %d channels x %d B/call (20 Msps x 52.4 ms)" % (T, M, nch, n),
        "ms_per_call": round(ms, 4), "iq_msamples_per_s": round(samples / ms / 1e3, 1),
        "realtime_factor_vs_256x20Msps": round(samples / ms / 1e3 / (256 * 20.0), 1),
        "algorithmic_GBps": round(alg / ms / 1e6, 1), "hbm_frac_of_8TBps": round(alg / ms / 1e6 / 8000.0, 4),
        "int_mac_per_s": round(macs / ms * 1e3, 0), "form": "valu" if os.environ.get("FMD_FIR_MFMA") == "0" else "mfma (v_mfma_i32_16x16x64_i8)",
        "outputs_per_channel": int(nout)}))


if __name__ == "__main__":
    main()
