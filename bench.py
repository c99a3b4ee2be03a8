#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s demodulated by the fused HIP path, with its HBM roofline and a CPU baseline.

One "step" = one call of fmd_demod_demodulate_device over one batch of synthetic IQ already
resident in HBM: BASELINE.json configs[2], 4096 FM channels x 262144 B (DEFAULT_BUF_LENGTH,
src/lib.rs:25) at the 2.4 Msps configuration (downsample 10, 240 kHz -> 32 kHz), per GPU.
Channels are independent (one Demod per stream, examples/simple_fm.rs:137), so N GPUs = N x 4096
channels with no data-path collective (weak scaling).  Prints ONE JSON line on rank 0.

Launching:  `python bench.py --gpus N` starts N rank processes ITSELF (fresh children, created before this
process touches the GPU; the parent only relays rank 0's line);  under `python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N` (RANK / WORLD_SIZE set) each process is one rank.  Both ways: one rank
per GPU, RCCL ("nccl") only for the barrier / max-over-ranks timing.  `--force-dist` takes that same
torch.distributed code path with world size 1 (so the RCCL branch runs on a one-GPU box too).

Timing: exactly K steps per timed region, each region bracketed by barrier + synchronize, MAX over ranks;
the region is repeated until >= 1 s has been timed and the MEDIAN region is reported (min / max beside it).
After the timing, at EVERY world size: a parity bit on every rank (a fresh bank, two steps from the zero state, >= 64
channels of the rank's own shard against the oracle; the flags are gathered into `parity`), every rank's shader clock
and socket power while all GPUs run the headline launch together (`extra.power_clock.per_gpu`), the name of the kernel
the library reports it launched (`roofline.kernel`, checked against the expected one) and, on rank 0, the CPU baseline
(the oracle, the only place besides the parity bit where bench.py touches it).  With one GPU also the side lines for the
other BASELINE configurations and the rest of the supported domain (`extra.*`, each outside the headline's timed region).
"""
import argparse
import ctypes as C
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHANNELS = 4096
BLOCK = 16 * 16384
D, FAST, SLOW = 10, 240000, 32000
CFG_REF = (6, 170000, 32000)     # optimal_settings(94.9 MHz, 170 kHz), examples/simple_fm.rs:25-27,48,189-214
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# the kernel this workload is expected to run (rocprofv3 --kernel-trace name); the line quotes what the library REPORTS it
# launched (fmd_demod_last_kernel) and flags a difference
KERNEL_EXPECTED = "fmd_tk::fmd_demod_tile_kernel<5, 2>"
MIN_TIMED_S = 1.0                # repeat the K-step region until this much has been timed
MAX_REGIONS = 4000
PMC_SUMMARY = os.path.join("profiles", "r06_pmc_summary.json")
BOUNDS = os.path.join("profiles", "r06_bounds.json")   # per-row pipe utilisation from the committed PMC passes (scripts/summarize_bounds.py)


# What each kernel family is built from.  Every counter-derived figure this file quotes (roofline.traffic, the pipe fractions of the
# `extra` rows) is tied to the hash of ITS family's sources as they were on the GPU box when the counters were taken: an edit of the FIR
# kernels no longer leaves the FIR rows quoting old counters (only the headline was tied), and an edit of one family
# does not void the others.  tests/test_isa_invariants.py checks the committed summaries against these hashes on the CPU box.
_TILE_COMMON = ("fmd_tile_body.h", "fmd_tile_launch.hip", "fmd_kernels.h", "fmd_device.h", "fmd_index.h", "fmd_host.h", "fmd_internal.h", "fmd_api.cpp")
KERNEL_FAMILIES = {
    "tile_even": ("fmd_tile_lds_even.hip",) + _TILE_COMMON,      # downsample 2 ... 14 (the headline: downsample 10)
    "tile_wide": ("fmd_tile_lds_wide.hip",) + _TILE_COMMON,      # downsample 16 ... 128 and the catch-all
    "tile_odd": ("fmd_tile_lds_odd.hip",) + _TILE_COMMON,        # odd factors
    "stream": ("fmd_tile_stream.hip",) + _TILE_COMMON,           # register-streaming kernels (downsample 2, 4)
    "fir": ("fmd_fir.hip", "fmd_fir_common.h", "fmd_host.h", "fmd_internal.h"),
    "fused": ("fmd_firdemod.hip", "fmd_fir_common.h", "fmd_device.h", "fmd_index.h", "fmd_kernels.h", "fmd_host.h", "fmd_internal.h"),
}
HEADLINE_FAMILY = "tile_even"


def family_hashes():
    """sha256 (16 hex digits) over each kernel family's sources."""
    csrc = os.path.join(ROOT, "rtl-sdr-rs_amd", "csrc")
    out = {}
    for fam, files in KERNEL_FAMILIES.items():
        h = hashlib.sha256()
        for f in files:
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
        out[fam] = h.hexdigest()[:16]
    return out


def kernel_source_hash():
    """The headline kernel's family: ties a committed PMC summary to the code it was measured on."""
    return family_hashes()[HEADLINE_FAMILY]


def family_of_kernel(name):
    """Kernel name (as fmd_*_last_kernel / rocprofv3 print it) -> family key."""
    import re
    name = name or ""
    if "fmd_firdemod" in name:
        return "fused"
    if "fmd_fir_" in name:
        return "fir"
    if "fmd_demod_stream_kernel" in name:
        return "stream"
    m = re.search(r"fmd_demod_tile_kernel<(-?\d+),", name)
    if m:
        dh = int(m.group(1))
        return "tile_odd" if dh < 0 else ("tile_even" if 1 <= dh <= 7 else "tile_wide")
    return None


def load_bounds():
    """profiles/r06_bounds.json (scripts/summarize_bounds.py): how busy the vector / scalar / matrix pipes were per row, from the committed
    PMC passes -- each row quoted only when it was taken on THESE sources of its kernel family."""
    try:
        with open(os.path.join(ROOT, BOUNDS)) as f:
            b = json.load(f)
    except (OSError, ValueError):
        return None
    now = family_hashes()
    meas = b.get("family_sha16") or {}
    b["family_current"] = {fam: meas.get(fam) == h for fam, h in now.items()}
    b["current"] = all(b["family_current"].values())
    return b


CO_BOUND_WITHIN = 0.03


def bound_fields(b, key, hbm_frac, section="demod"):
    """`bound` + the pipe fractions of one row: the largest of the launch's HBM fraction and the vector / scalar / matrix pipes' issue
    fractions under the counters -- "co-bound" when the two largest are within 0.03 of each other (the model is good to a few
    percent: scripts/summarize_bounds.py, profiles/README.md).  A row whose kernel family has been edited since the counters were
    taken is REFUSED (no figures), not quoted as "indicative"."""
    if not b:
        return {}
    row = (b.get(section) or {}).get(key) if section == "demod" else b.get(section)
    if not row:
        return {}
    fam = family_of_kernel(row.get("kernel")) or {"config4_fir": "fir", "config4_fir_demod_fused": "fused"}.get(section)
    if not (b.get("family_current") or {}).get(fam, False):
        return {"bound_source": "%s: refused -- its counters were taken on other sources of the '%s' kernels; re-run scripts/gpu_round.sh" % (BOUNDS, fam)}
    out = {k: row[k] for k in ("valu_issue_frac", "salu_issue_frac", "mfma_busy_frac", "lds_bank_conflict_share", "compute_alone_cycles_frac") if row.get(k) is not None}
    cand = {"hbm": hbm_frac, "valu": row.get("valu_issue_frac") or 0.0, "salu": row.get("salu_issue_frac") or 0.0, "mfma": row.get("mfma_busy_frac") or 0.0}
    order = sorted(cand, key=cand.get, reverse=True)
    out["bound"] = "co-bound" if cand[order[0]] - cand[order[1]] <= CO_BOUND_WITHIN else order[0]
    out["bound_top2"] = order[:2]
    out["bound_source"] = BOUNDS
    return out


def load_oracle():
    """The CPU oracle (oracle/fm_oracle.c): checker and CPU baseline only -- never on the measured path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    return oracle_lib.load()


def cpu_baseline(fmd, torch, cfg, iq_dev, target_s=4.0):
    """The oracle (oracle/fm_oracle.c, a C restatement of the reference passes: kind 'port') timed on this
    box's host cores over a bounded sample of the same workload; BASELINE.md's CPU-1 line (the reference's own
    configuration, one channel, one thread, >= 8 blocks) beside it."""
    import numpy as np
    o = load_oracle()
    cores = os.cpu_count() or 1
    ocfg = o.config(cfg.downsample, cfg.rate_out, cfg.rate_resample)
    chans = min(CHANNELS, cores * 2)
    calls = 4                                                 # consecutive blocks per channel (state carries)
    host = iq_dev[:chans].cpu().numpy()                       # [chans][BLOCK] of the GPU run's own input
    data = np.ascontiguousarray(np.tile(host[:, None, :], (1, calls, 1)).reshape(chans, calls * BLOCK))
    u8p = C.POINTER(C.c_uint8)
    chk = C.c_uint64()

    def run():
        return o.lib.fmo_bench_batch(C.byref(ocfg), data.ctypes.data_as(u8p), chans, calls, BLOCK, cores,
                                     C.byref(chk), None)
    if run() <= 0:                                            # warm-up pass (page faults, thread start)
        return None
    total, passes = 0.0, 0
    while total < target_s and passes < 4000:                 # ~4 s of CPU work by default (the all-threads figure converges in < 2 s), bounded
        t = run()
        if t <= 0:
            return None
        total += t
        passes += 1
    samples = passes * chans * calls * (BLOCK // 2)

    def one_thread(c, blocks, budget_s):
        """one thread, one channel, `blocks` consecutive 262144-byte calls; returns Msamples/s"""
        d1 = np.ascontiguousarray(np.tile(host[0], blocks))
        t_sum, n = 0.0, 0
        while t_sum < budget_s and n < 400:
            t = o.lib.fmo_bench_batch(C.byref(c), d1.ctypes.data_as(u8p), 1, blocks, BLOCK, 1, C.byref(chk), None)
            if t <= 0:
                return None
            t_sum += t
            n += 1
        return round(n * blocks * (BLOCK // 2) / t_sum / 1e6, 2)

    single = one_thread(ocfg, calls, 1.0)                     # one thread, one channel, same blocks
    ref_cfg = o.config(*CFG_REF)
    cpu1 = one_thread(ref_cfg, 8, 2.0)
    return {"value": round(samples / total / 1e6, 2), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "single_thread_value": single,
            "cfg_ref_single_thread": {"value": cpu1, "unit": "Msamples/s", "cores": 1,
                                      "ns_per_iq_sample": round(1e3 / cpu1, 3) if cpu1 else None,
                                      "what": "BASELINE.md CPU-1: downsample %d, %d -> %d Hz, 1 channel, 1 thread, 8 x %d B per pass" % (CFG_REF + (BLOCK,))},
            "sample": "%d passes x %d channels x %d calls x %d B of the same synthetic workload, %d threads, "
                      "%.1f s of oracle time (oracle/fm_oracle.c, gcc -O3 -fwrapv)" % (passes, chans, calls, BLOCK, cores, total)}


def parity_bit(fmd, torch, cfg, bufs, dev_index, stream, n_check=64):
    """Outside every timed region: a FRESH bank (zero state) runs two consecutive steps on the benchmark's own
    input batches; audio of >= 64 channels (first, last, strided) and the final state are compared with the oracle."""
    import numpy as np
    o = load_oracle()
    nch = bufs[0].shape[0]
    picks = sorted(set([0, 1, nch - 1, nch // 2] + list(range(0, nch, max(1, nch // (n_check - 4))))))
    bank = fmd.DemodBank(cfg, nch, device_id=dev_index)
    cap = bank.out_cap(BLOCK)
    obank = o.new_bank(o.config(cfg.downsample, cfg.rate_out, cfg.rate_resample), len(picks))
    ok, bad = True, []
    for call in range(2):
        out = torch.zeros((nch, cap), dtype=torch.int16, device=bufs[0].device)
        src = bufs[call % len(bufs)]
        bank.demodulate_device(src.data_ptr(), BLOCK, out.data_ptr(), cap, None, stream)
        bank.check()
        lens = bank.last_out_len()
        got = out[picks].cpu().numpy()
        exp, elens = o.demodulate_batch(obank, src[picks].cpu().numpy())
        for j, c in enumerate(picks):
            if lens[c] != elens[j] or not np.array_equal(got[j, :elens[j]], exp[j, :elens[j]]):
                ok = False
                bad.append([call, int(c)])
    for j in (0, len(picks) - 1):
        if bank.get_state(picks[j]).as_dict() != o.state_of(obank[j]):
            ok = False
            bad.append(["state", int(picks[j])])
    bank.close()
    return {"channels_checked": len(picks), "calls": 2, "ok": ok, "mismatches": bad[:8],
            "against": "oracle/fm_oracle.c on the same bytes, after the timed regions, fresh bank from the zero state"}


def time_calls(torch, call, settle=150, steps=100, regions=5):
    """Median / min / max ms per call over `regions` HIP-event-timed regions of `steps` calls on torch's current
    stream (the stream every call here is enqueued on), after `settle` untimed calls."""
    for i in range(settle):
        call(i)
    torch.cuda.synchronize()
    regs, last = [], None
    for _ in range(regions):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            last = call(i)
        e1.record()
        torch.cuda.synchronize()
        regs.append(e0.elapsed_time(e1) / steps)
    regs.sort()
    return regs[len(regs) // 2], regs[0], regs[-1], last


def extra_config4(fmd, torch, dev, stream, fused, bounds=None):
    """BASELINE configs[3] beside the headline (outside its timed region): 127-tap FIR, decimate 8, 256 channels x
    2 MiB per call.  fused=False: the stand-alone operator fmd_fir_* (complex i32 out, 3 B per IQ sample);
    fused=True: FIR -> discriminator -> resampler in one kernel (fmd_firdemod_*, s16 audio out)."""
    import numpy as np
    nch, n, T, M = 256, 2 << 20, 127, 8
    rng = np.random.default_rng(1)
    bufs = []
    for b in range(3):
        t = torch.empty((nch, n), dtype=torch.uint8, device=dev)
        fmd.synth.fill_device(t.data_ptr(), nch, n, sample_offset=b * (n // 2), device_id=dev.index, stream=stream)
        bufs.append(t)
    taps = rng.integers(-2047, 2048, T).astype(np.int16)             # the same taps for both lines
    if fused:
        fast, slow = 2_500_000, 48_000                               # 20 Msps / 8 -> 2.5 Msps -> 48 kHz audio; auto shift
        bank = fmd.FirDemodBank(taps, M, fast, slow, nch, device_id=dev.index)
        cap = bank.out_cap(n)
        out = torch.zeros((nch, cap), dtype=torch.int16, device=dev)
        call = lambda i: bank.demodulate_device(bufs[i % 3].data_ptr(), n, out.data_ptr(), cap, stream)
        out_bytes = lambda k: 2 * k
    else:
        bank = fmd.FirBank(taps, M, nch, device_id=dev.index)
        cap = bank.out_cap(n)
        out = torch.zeros((nch, cap, 2), dtype=torch.int32, device=dev)
        # A third of this operator's bytes are WRITTEN (int32 re, im), and one 268 MB output buffer written again by every call partly
        # stays in the 256 MB memory-side cache.  `frac` is therefore measured with the calls rotating over FOUR output buffers (1.07 GB:
        # every written byte goes to HBM) -- the steady state of a consumer that keeps its outputs (the honest figure);
        # the one-buffer run is the side field `one_output_buffer`.
        outs = [out] + [torch.zeros_like(out) for _ in range(3)]
        call = lambda i: bank.filter_device(bufs[i % 3].data_ptr(), n, outs[i % 4].data_ptr(), cap, stream)
        out_bytes = lambda k: 8 * k
    ms, lo, hi, nout = time_calls(torch, call)
    if fused:
        bank.check()
    alg = nch * n + nch * out_bytes(int(nout))
    res = {"workload": "BASELINE configs[3]: %d-tap FIR, decimate %d, %d channels x %d B/call (20 Msps x 52.4 ms)%s"
                       % (T, M, nch, n, ", fused with the discriminator and the %d -> %d Hz resampler" % (fast, slow) if fused else ""),
           "kernel": bank.kernel_name(), "ms_per_call": round(ms, 4),
           "ms_min_max": [round(lo, 4), round(hi, 4)],
           "iq_msamples_per_s": round(nch * (n // 2) / ms / 1e3, 1), "outputs_per_channel": int(nout),
           "algorithmic_bytes_per_launch": alg, "GBps": round(alg / ms / 1e6, 1), "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4)}
    res.update(bound_fields(bounds, None, res["frac"], section="config4_fir_demod_fused" if fused else "config4_fir"))
    if fused:
        # the same shape with an 8-bit filter (every |tap| <= 127): one i8 digit per tap, (re, im) of eight outputs per operand fragment,
        # 24 instead of 40 matrix instructions per wave (fmd_firdemod_reg1_kernel) -- here, where the outputs are
        # 0.3 % of the bytes and the matrix phase is the largest region, it pays
        taps8 = rng.integers(-127, 128, T).astype(np.int16)
        bank8 = fmd.FirDemodBank(taps8, M, fast, slow, nch, device_id=dev.index)
        ms8, lo8, hi8, _ = time_calls(torch, lambda i: bank8.demodulate_device(bufs[i % 3].data_ptr(), n, out.data_ptr(), cap, stream),
                                      settle=60, steps=60, regions=3)
        bank8.check()
        res["taps_8bit_one_digit"] = {"kernel": bank8.kernel_name(), "ms_per_call": round(ms8, 4), "frac": round(alg / ms8 / 1e6 / HBM_PEAK_GBS, 4)}
        del bank8
    if not fused:
        ms1, lo1, hi1, _ = time_calls(torch, lambda i: bank.filter_device(bufs[i % 3].data_ptr(), n, out.data_ptr(), cap, stream),
                                      settle=60, steps=60, regions=3)
        res["output_buffers"] = 4
        res["one_output_buffer"] = {"ms_per_call": round(ms1, 4), "frac": round(alg / ms1 / 1e6 / HBM_PEAK_GBS, 4),
                                    "note": "the same calls writing ONE 268 MB buffer again and again: partly resident in the 256 MB memory-side cache (round 5's `frac`)"}
        res["note"] = ("33 % of the bytes are outputs; reads alone stream at ~6.7 TB/s, written bytes at ~4.9 TB/s, the matrix phase is "
                       "hidden (-1.9 % without it): profiles/r05_experiments.md section 12; round 6: outputs leave the wave in lane order "
                       "(whole lines per store instruction): profiles/r06_experiments.md")
        del outs
        # the same shape with an 8-bit filter (every |tap| <= 127): one i8 digit per tap, eight outputs per operand column -- the
        # same matrix instructions and LDS operand reads cover twice the outputs
        taps8 = rng.integers(-127, 128, T).astype(np.int16)
        bank8 = fmd.FirBank(taps8, M, nch, device_id=dev.index)
        outs8 = [out] + [torch.zeros_like(out) for _ in range(3)]
        ms8, lo8, hi8, _ = time_calls(torch, lambda i: bank8.filter_device(bufs[i % 3].data_ptr(), n, outs8[i % 4].data_ptr(), cap, stream),
                                      settle=60, steps=60, regions=3)
        del outs8
        res["taps_8bit_one_digit"] = {"tap_digits": bank8.tap_digits(), "kernel": bank8.kernel_name(), "ms_per_call": round(ms8, 4), "frac": round(alg / ms8 / 1e6 / HBM_PEAK_GBS, 4)}
        res["tap_digits"] = bank.tap_digits()
        del bank8
    del bank, out, bufs
    return res


def extra_cfg_ref(fmd, torch, dev, stream, bufs, bounds=None):
    """The reference's OWN configuration -- what optimal_settings(94.9 MHz, 170 kHz) produces for the shipped example
    (simple_fm.rs:25-27,48: downsample 6, 170 kHz -> 32 kHz) -- on the headline's batch shape and input buffers."""
    d, fast, slow = CFG_REF
    nch = bufs[0].shape[0]
    cfg = fmd.DemodConfig(fast, fast, slow, d, max(1, (1 << 15) // (128 * d)))
    bank = fmd.DemodBank(cfg, nch, device_id=dev.index)
    cap = bank.out_cap(BLOCK)
    out = torch.zeros((nch, cap), dtype=torch.int16, device=dev)
    call = lambda i: bank.demodulate_device(bufs[i % len(bufs)].data_ptr(), BLOCK, out.data_ptr(), cap, None, stream)
    ms, lo, hi, _ = time_calls(torch, call)
    bank.check()
    alg = nch * BLOCK + 2 * int(bank.last_out_len().sum())
    res = {"workload": "cfg-ref: %d channels x %d B/call, downsample %d, %d -> %d Hz (examples/simple_fm.rs:25-27)" % (nch, BLOCK, d, fast, slow),
           "kernel": bank.last_kernel(), "ms_per_call": round(ms, 4), "ms_min_max": [round(lo, 4), round(hi, 4)],
           "iq_msamples_per_s": round(nch * (BLOCK // 2) / ms / 1e3, 1), "algorithmic_bytes_per_launch": alg,
           "GBps": round(alg / ms / 1e6, 1), "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4), "tiling": bank.tiling()}
    res.update(bound_fields(bounds, "%d,%d,%d" % (d, fast, slow), res["frac"]))
    bank.close()
    return res


# (downsample, rate_out, rate_resample): the rest of the supported domain, tools/bench_configs.py's table
DOMAIN = [(1, 48000, 48000), (2, 500000, 32000), (3, 400000, 48000), (4, 256000, 48000), (5, 250000, 44100), (7, 166666, 32000), (8, 250000, 44100),
          (12, 192000, 32000), (16, 150000, 32000), (64, 37500, 8000)]


def extra_domain(fmd, torch, dev, stream, bufs, bounds=None):
    """Every other kernel family of the demod path on the headline's batch shape and input buffers (outside its timed
    region, <= 0.2 s each): odd factors, the register-streaming kernel (2, 4), the wrap-around walk (16, 64), one
    discriminator per IQ sample (1).  Each line names the kernel the library reports it launched."""
    nch = bufs[0].shape[0]
    rows = []
    for d, fast, slow in DOMAIN:
        try:
            cfg = fmd.DemodConfig(fast, fast, slow, d, max(1, (1 << 15) // (128 * d)))
            bank = fmd.DemodBank(cfg, nch, device_id=dev.index)
            cap = bank.out_cap(BLOCK)
            out = torch.zeros((nch, cap), dtype=torch.int16, device=dev)
            call = lambda i: bank.demodulate_device(bufs[i % len(bufs)].data_ptr(), BLOCK, out.data_ptr(), cap, None, stream)
            ms, lo, hi, _ = time_calls(torch, call, settle=60, steps=40, regions=3)
            bank.check()
            alg = nch * BLOCK + 2 * int(bank.last_out_len().sum())
            row = {"downsample": d, "rate_out": fast, "rate_resample": slow, "kernel": bank.last_kernel(),
                   "ms_per_call": round(ms, 4), "ms_min_max": [round(lo, 4), round(hi, 4)], "GBps": round(alg / ms / 1e6, 1),
                   "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4), "audio_per_tile": bank.tiling()["audio_per_tile"]}
            row.update(bound_fields(bounds, "%d,%d,%d" % (d, fast, slow), row["frac"]))
            rows.append(row)
            bank.close()
            del out
        except Exception as e:
            rows.append({"downsample": d, "rate_out": fast, "rate_resample": slow, "error": repr(e)})
    return {"workload": "%d channels x %d B/call per configuration, 3 regions of 40 calls after 60 untimed ones" % (nch, BLOCK), "rows": rows}


def extra_skeleton(fmd, torch, dev, stream, bufs):
    """The kernels' own ceiling on THIS box: the staging skeleton of the tile kernel -- prologue, LDS-DMAs, barrier, one store; no
    rounds, no resampler (ablation bit 3 of the -DFMD_EXPERIMENT build, loaded side by side here; the shipped library has no such
    switch) -- on the headline's batch and on the reference's own rates.  `frac_of_skeleton` of a row = how much of the time its
    memory side alone would need the whole kernel takes; the skeleton itself runs at the full shader clock, the kernels at the
    power-capped one (extra.power_clock)."""
    from rtl_sdr_rs_amd import _ffi
    exp = os.path.join(ROOT, "rtl-sdr-rs_amd", "libfmd_hip_exp.so")
    if not os.path.exists(exp):
        return {"error": "libfmd_hip_exp.so not built"}
    lib = C.CDLL(exp)
    for name, (res_t, args) in _ffi.PROTOTYPES.items():
        if hasattr(lib, name):
            getattr(lib, name).restype, getattr(lib, name).argtypes = res_t, args
    nch = bufs[0].shape[0]
    out = {"what": "fmd_demod_tile_kernel with FMD_DBG=8 (experiment build): staging only; %d channels x %d B per launch" % (nch, BLOCK)}
    saved = os.environ.get("FMD_DBG")
    try:
        for key, (d, fast, slow) in (("headline", (D, FAST, SLOW)), ("cfg_ref", CFG_REF)):
            cfg = fmd.DemodConfig(fast, fast, slow, d, max(1, (1 << 15) // (128 * d)))
            cap = int(lib.fmd_out_cap(C.byref(cfg), BLOCK))
            o = torch.zeros((nch, cap), dtype=torch.int16, device=dev)
            os.environ["FMD_DBG"] = "8"                          # read when the handle is created (experiment build only)
            h = C.c_void_p()
            dc = fmd.DeviceConfig(nch, dev.index, 0)
            if lib.fmd_demod_new(C.byref(cfg), C.byref(dc), C.byref(h)) != 0:
                out[key] = {"error": "fmd_demod_new failed"}
                continue
            call = lambda i: lib.fmd_demod_demodulate_device(h, bufs[i % len(bufs)].data_ptr(), BLOCK, o.data_ptr(), cap, None, stream)
            ms, lo, hi, _ = time_calls(torch, call, settle=60, steps=60, regions=3)
            lib.fmd_demod_check(h)
            lib.fmd_demod_free(h)
            out[key] = {"ms_per_call": round(ms, 4), "ms_min_max": [round(lo, 4), round(hi, 4)], "input_GBps": round(nch * BLOCK / ms / 1e6, 1),
                        "frac_of_spec_reads_only": round(nch * BLOCK / ms / 1e6 / HBM_PEAK_GBS, 4)}
            del o
    finally:
        if saved is None:
            os.environ.pop("FMD_DBG", None)
        else:
            os.environ["FMD_DBG"] = saved
    return out


def extra_config2(fmd, torch, dev, stream):
    """BASELINE configs[1]: ONE FM channel at 2.4 Msps: (a) one DEFAULT_BUF_LENGTH call per launch (launch-latency bound);
    (b) 64 MiB per launch handed over as 256 reference calls of 262144 B (fmd_demod_set_block_len): the audio and the
    state are those of feeding the reference the 256 blocks one by one."""
    cfg = fmd.DemodConfig(FAST, FAST, SLOW, D, max(1, (1 << 15) // (128 * D)))
    res = {"workload": "BASELINE configs[1]: 1 FM channel @ 2.4 Msps, device-resident input"}
    for name, n, blk, steps in (("one_262144_B_call_per_launch", BLOCK, 0, 400), ("64MiB_per_launch_as_256_reference_calls", 64 << 20, BLOCK, 100)):
        bank = fmd.DemodBank(cfg, 1, device_id=dev.index)
        bank.set_block_len(blk)
        iq = torch.empty((1, n), dtype=torch.uint8, device=dev)
        fmd.synth.fill_device(iq.data_ptr(), 1, n, device_id=dev.index, stream=stream)
        cap = bank.out_cap(n)
        out = torch.zeros((1, cap), dtype=torch.int16, device=dev)
        call = lambda i: bank.demodulate_device(iq.data_ptr(), n, out.data_ptr(), cap, None, stream)
        ms, lo, hi, _ = time_calls(torch, call, settle=50, steps=steps, regions=3)
        bank.check()
        res[name] = {"ms_per_launch": round(ms, 5), "ms_min_max": [round(lo, 5), round(hi, 5)],
                     "iq_msamples_per_s": round(n / 2 / ms / 1e3, 1), "realtime_factor_at_2.4Msps": round(n / 2 / ms / 1e3 / 2.4, 1),
                     "GBps": round(n / ms / 1e6, 1)}
        bank.close()
        del iq, out
    return res


def extra_check_per_step(fmd, torch, bank, bufs, out, cap, stream, steps=200):
    """The drop-in usage of the device entry point: fmd_demod_demodulate_device followed by fmd_demod_check (sync +
    report read-back, the completion point a consumer needs before it reads the audio) for EVERY buffer.  Host wall
    time per step, launch latency and the synchronisation included."""
    for i in range(20):
        bank.demodulate_device(bufs[i % len(bufs)].data_ptr(), BLOCK, out.data_ptr(), cap, None, stream)
        bank.check()
    t0 = time.perf_counter()
    for i in range(steps):
        bank.demodulate_device(bufs[i % len(bufs)].data_ptr(), BLOCK, out.data_ptr(), cap, None, stream)
        bank.check()
    ms = (time.perf_counter() - t0) / steps * 1e3
    nch = bufs[0].shape[0]
    return {"what": "demodulate_device + fmd_demod_check after every step (host wall time, %d steps)" % steps,
            "ms_per_step": round(ms, 4), "iq_msamples_per_s": round(nch * (BLOCK // 2) / ms / 1e3, 1)}


def extra_check_pipelined(fmd, torch, bank, bufs, out, cap, stream, steps=200):
    """The same cadence with the completion point TWO LAUNCHES BACK (fmd_demod_check_behind, round 6): enqueue buffer n, settle buffer
    n - 2 while n - 1 and n run -- what `loop { demodulate(buf); output(audio) }` of simple_fm.rs:150-156 becomes when the consumer lags
    two buffers; a whole launch stays queued behind the running one, so a late host costs nothing.  Three output buffers rotate (a
    launch's buffer stays untouched until it is settled).  Host wall time per step; `one_launch_back`: the same with back = 1."""
    outs = (out, torch.zeros_like(out), torch.zeros_like(out))

    def run(back):
        for i in range(20):
            bank.demodulate_device(bufs[i % len(bufs)].data_ptr(), BLOCK, outs[i % 3].data_ptr(), cap, None, stream)
            bank.check_behind(back)
        bank.check()
        t_enq = t_wait = 0.0
        t0 = time.perf_counter()
        for i in range(steps):
            ta = time.perf_counter()
            bank.demodulate_device(bufs[i % len(bufs)].data_ptr(), BLOCK, outs[i % 3].data_ptr(), cap, None, stream)
            tb = time.perf_counter()
            bank.check_behind(back)
            tc = time.perf_counter()
            t_enq += tb - ta; t_wait += tc - tb
        bank.check()
        ms = (time.perf_counter() - t0) / steps * 1e3
        return ms, t_enq / steps * 1e3, t_wait / steps * 1e3

    g0 = bank.f64_stats()["guarded"]
    ms, enq, wait = sorted(run(2) for _ in range(3))[1]          # median of three runs of `steps` steps: one host hiccup of a millisecond is 3 % of a run
    guarded = bank.f64_stats()["guarded"] - g0       # launches with a report are settled while the newer ones run (no drain)
    ms1, _, _ = sorted(run(1) for _ in range(3))[1]
    nch = bufs[0].shape[0]
    return {"what": "demodulate_device(n) + fmd_demod_check_behind(2) (settles n - 2 while n - 1 and n run) every step, fmd_demod_check at the end (host wall time, median of 3 runs of %d steps)" % steps,
            "ms_per_step": round(ms, 4), "iq_msamples_per_s": round(nch * (BLOCK // 2) / ms / 1e3, 1),
            "host_ms_in_enqueue": round(enq, 4), "host_ms_in_check_behind": round(wait, 4), "guarded_samples_settled_in_flight": int(guarded),
            "one_launch_back": {"ms_per_step": round(ms1, 4), "what": "the same with fmd_demod_check_prev (back = 1): one launch of look-ahead"}}


def extra_sink_pcie(fmd, dev_index, nch=1024, steps=20):
    """PCIe-INCLUSIVE rate (never `value`): 1024 read_sync-sized buffers per slot through the pipelined sink (fmd_sink_*,
    ring of 3 page-locked slots; H2D, kernel and D2H overlap), with and without the memcpy that fills the slot."""
    import numpy as np
    from rtl_sdr_rs_amd._ffi import SINK_CALLBACK, check, lib
    cfg = fmd.DemodConfig(FAST, FAST, SLOW, D, max(1, (1 << 15) // (128 * D)))
    src = np.ascontiguousarray(np.tile(fmd.synth.synth_iq(8, BLOCK), (nch // 8, 1)))
    count = [0, 0]

    def _count(user, seq, audio, out_len, out_cap, status):
        count[0] += 1
        count[1] |= int(status != 0)
    cb = SINK_CALLBACK(_count)
    h = C.c_void_p()
    ids = (C.c_int32 * 1)(dev_index)
    check(lib().fmd_sink_new(C.byref(cfg), nch, ids, 1, BLOCK, 3, C.cast(cb, C.c_void_p), None, C.byref(h)))

    def push(fill):
        p = C.c_void_p()
        check(lib().fmd_sink_acquire(h, C.byref(p)))
        if fill:
            C.memmove(p.value, src.ctypes.data, nch * BLOCK)     # stands for read_sync writing into the slot
        check(lib().fmd_sink_submit(h))
    res = {"what": "%d channels x %d B per buffer through fmd_sink_* (depth 3), host buffers: H2D + kernel + D2H" % (nch, BLOCK)}
    try:
        for _ in range(4):
            push(True)
        check(lib().fmd_sink_drain(h))
        for name, fill in (("slots_filled_by_memcpy", True), ("slots_prefilled", False)):
            t0 = time.perf_counter()
            for _ in range(steps):
                push(fill)
            check(lib().fmd_sink_drain(h))
            dt = (time.perf_counter() - t0) / steps
            res[name] = {"ms_per_buffer": round(dt * 1e3, 3), "host_GBps": round(nch * BLOCK / dt / 1e9, 2),
                         "iq_msamples_per_s": round(nch * BLOCK / 2 / dt / 1e6, 1)}
        res["delivered"] = count[0]
        res["all_status_ok"] = count[1] == 0
    finally:
        lib().fmd_sink_free(h)
    return res


class GpuSensors:
    """Shader clock and socket power of ONE device from amdgpu's sysfs files (readable by an ordinary user): the
    device is matched by its PCI address, because a box lists every GPU of the node under /sys/class/drm."""

    def __init__(self, torch, dev_index):
        self.dir = None
        try:
            p = torch.cuda.get_device_properties(dev_index)
            d = "/sys/bus/pci/devices/%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
            if os.path.exists(os.path.join(d, "pp_dpm_sclk")):
                self.dir = d
        except Exception:
            pass

    def _hwmon(self, name):
        import glob
        for f in glob.glob(os.path.join(self.dir, "hwmon", "hwmon*", name)):
            try:
                return int(open(f).read().strip())
            except (OSError, ValueError):
                pass
        return None

    def sclk_mhz(self):
        import re
        try:
            lines = open(os.path.join(self.dir, "pp_dpm_sclk")).read().splitlines()
        except OSError:
            return None
        cur = [l for l in lines if l.strip().endswith("*")]
        m = re.search(r"(\d+)\s*[Mm][Hh]z", cur[0]) if cur else None
        return int(m.group(1)) if m else None

    def sclk_max_mhz(self):
        import re
        try:
            v = [int(x) for x in re.findall(r"(\d+)\s*[Mm][Hh]z", open(os.path.join(self.dir, "pp_dpm_sclk")).read())]
        except OSError:
            return None
        return max(v) if v else None

    def power_w(self):
        for name in ("power1_average", "power1_input"):
            v = self._hwmon(name)
            if v is not None:
                return v / 1e6
        return None

    def power_cap_w(self):
        v = self._hwmon("power1_cap")
        return v / 1e6 if v is not None else None

    def sample_while(self, work, seconds, period=0.05):
        """Run work() repeatedly for `seconds` while a thread samples clock and power; returns the sample lists."""
        import threading
        sclk, power, stop = [], [], []

        def loop():
            while not stop:
                s, p = self.sclk_mhz(), self.power_w()
                if s is not None:
                    sclk.append(s)
                if p is not None:
                    power.append(p)
                time.sleep(period)

        th = threading.Thread(target=loop, daemon=True)
        th.start()
        try:
            t_end = time.time() + seconds
            while time.time() < t_end:
                work()
        finally:                                             # the sampler ends with the work, also when the work raises
            stop.append(1)
            th.join()
        return sclk, power


def sensor_stats(v, nd=0):
    if not v:
        return None
    v = sorted(v)
    return {"n": len(v), "min": round(v[0], nd), "median": round(v[len(v) // 2], nd), "max": round(v[-1], nd)}


def extra_power_clock(torch, dev_index, step, seconds=2.0):
    """Shader clock and socket power while the headline launch runs back to back (outside the timed regions), beside an
    idle reading.  Why it is in the line: on the boxes measured so far the full kernel sits AT the board's power cap
    and the firmware lowers the shader clock by 10-15 % (the staging skeleton and the compute side alone each run at
    the full clock, profiles/r03_experiments.md section 11), which is what separates the kernel from its own
    loads-only time."""
    sens = GpuSensors(torch, dev_index)
    if sens.dir is None:
        return {"error": "no amdgpu sysfs sensors for this device"}
    torch.cuda.synchronize()
    time.sleep(0.5)
    idle_s, idle_p = sens.sample_while(lambda: time.sleep(0.05), 0.5)
    ms = []

    def work():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(500):
            step(i)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / 500)

    work()                                                   # ramp
    ms.clear()
    sclk, power = sens.sample_while(work, seconds)
    cap, ps = sens.power_cap_w(), sensor_stats(power)
    res = {"what": "amdgpu sysfs (pp_dpm_sclk, hwmon power1_average) sampled every 50 ms during %.1f s of back-to-back headline launches" % seconds,
           "ms_per_call_during_sampling": round(sorted(ms)[len(ms) // 2], 4) if ms else None,
           "sclk_mhz": sensor_stats(sclk), "sclk_max_level_mhz": sens.sclk_max_mhz(), "power_w": ps, "power_cap_w": cap,
           "idle": {"sclk_mhz": sensor_stats(idle_s), "power_w": sensor_stats(idle_p)}}
    if cap and ps:
        res["at_power_cap"] = bool(ps["median"] >= 0.95 * cap)
    return res


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this process has not touched
    the GPU and never will), relay rank 0's JSON line, fail if any rank fails.  A rank that dies -- rank 0, which hosts
    the rendezvous store, included -- takes the job down after a grace period instead of leaving the others waiting in
    the rendezvous or a collective until their own timeout (10 to 30 minutes)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(args.gpus),
               LOCAL_WORLD_SIZE=str(args.gpus), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))

    def reap(grace_s):
        """let the others report their own error (or finish) first, then kill what is left -- by handle"""
        deadline = time.time() + grace_s
        while time.time() < deadline and any(p.poll() is None for p in procs):
            time.sleep(0.2)
        for p in procs:
            if p.poll() is None:
                p.kill()
    out0 = b""
    while True:
        try:
            out0, _ = procs[0].communicate(timeout=1.0)
            break                                                # rank 0 is done (whatever its exit code)
        except subprocess.TimeoutExpired:
            if any(p.poll() not in (None, 0) for p in procs[1:]):
                reap(15.0)
                out0, _ = procs[0].communicate()
                break
    # rank 0 failed: 15 s for the others' own messages; rank 0 succeeded: the others are past the last barrier and exit
    # by themselves -- a minute is generous
    reap(15.0 if procs[0].returncode != 0 else 60.0)
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    if any(rcs):
        sys.stderr.write("bench.py: rank exit codes %r\n" % (rcs,))
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--channels", type=int, default=CHANNELS)
    ap.add_argument("--nbuf", type=int, default=3, help="distinct input batches rotated through (defeats the 256 MiB L3)")
    ap.add_argument("--kt", type=int, default=0, help="tiling override (audio samples per tile)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline and the parity bit (both use the oracle)")
    ap.add_argument("--no-extra", action="store_true", help="skip the side lines (extra.*), the per-GPU clock / power sample included")
    ap.add_argument("--power-only", action="store_true", help="of the side lines keep only extra.power_clock (every rank samples its own GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="oracle time spent on the all-threads CPU baseline (rank 0); + 1 s + 2 s for the two single-thread figures")
    ap.add_argument("--min-timed-s", type=float, default=MIN_TIMED_S, help="repeat the K-step region until this much has been timed")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / max-over-ranks timing (nccl = RCCL); with gloo "
                         "ranks may share a GPU (launch-path check on a single-GPU box)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even with ONE rank: barrier / all_reduce / all_gather run through "
                         "the same code as the N-GPU job (exercises the RCCL branch on a one-GPU box)")
    ap.add_argument("--settle", type=int, default=150,
                    help="untimed steps before the W warm-up steps: the first ~10 ms after an idle period run at "
                         "ramping clocks (measured 0.21-0.23 ms/step vs 0.193 settled); reported in config")
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1:
        ap.error("--gpus and --steps must be >= 1")

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))                              # before torch / HIP are even imported
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus" % (args.gpus, world))

    import faulthandler
    faulthandler.enable()                                    # a rank that dies in native code leaves its Python stack on stderr
    import torch
    import rtl_sdr_rs_amd as fmd

    ndev = torch.cuda.device_count()
    if ndev < 1 or fmd.device_count() < 1:
        raise RuntimeError("bench.py: no gfx950 device; the product has no CPU path")
    if world > ndev and args.backend == "nccl":
        raise SystemExit("bench.py: --gpus %d but only %d device(s) visible (one rank per GPU; --backend gloo lets ranks "
                         "share a GPU for a launch-path check)" % (world, ndev))
    dist = None
    dev_index = local_rank % ndev
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                      # --force-dist without a launcher
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    cfg = fmd.DemodConfig(FAST, FAST, SLOW, D, max(1, (1 << 15) // (128 * D)))
    nch = args.channels
    lo, hi = fmd.shard.channel_range(nch * world, world, rank)          # this rank's global channel ids
    bank = fmd.DemodBank(cfg, nch, device_id=dev_index)
    if args.kt:
        bank.set_tiling(args.kt)
    cap = bank.out_cap(BLOCK)
    stream = torch.cuda.current_stream().cuda_stream
    bufs = []
    for b in range(args.nbuf):
        t = torch.empty((nch, BLOCK), dtype=torch.uint8, device=dev)
        fmd.synth.fill_device(t.data_ptr(), nch, BLOCK, sample_offset=b * (BLOCK // 2), device_id=dev_index,
                              stream=stream, seed=fmd.synth.DEFAULTS["seed"] + lo)
        bufs.append(t)
    out = torch.zeros((nch, cap), dtype=torch.int16, device=dev)

    def step(i):
        bank.demodulate_device(bufs[i % args.nbuf].data_ptr(), BLOCK, out.data_ptr(), cap, None, stream)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    for i in range(args.settle):
        step(i)
    for i in range(args.warmup):
        step(i)
    n_done = args.warmup

    def timed_region():
        """EXACTLY --steps steps, barrier + synchronize on both sides; returns (wall s [max over ranks], own wall s,
        HIP-event ms per step on the launch stream)."""
        nonlocal n_done
        fence()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for i in range(args.steps):
            step(n_done + i)
        ev1.record()
        fence()
        own = time.perf_counter() - t0
        n_done += args.steps
        bank.check()        # outside the bracket: settles and drains the guarded-f64 report buffer (1024 records: a minute of launches fills it)
        return max_over_ranks(own), own, ev0.elapsed_time(ev1) / args.steps

    regions = [timed_region()]
    # the same count on every rank (all-reduced time); 1.3: the first region of a short run is the slowest
    reps = min(MAX_REGIONS, max(1, int(math.ceil(1.3 * args.min_timed_s / regions[0][0]))))
    for _ in range(reps - 1):
        regions.append(timed_region())
    walls = sorted(r[0] for r in regions)
    elapsed = walls[len(walls) // 2]                                        # median region
    own_med = sorted(r[1] for r in regions)[len(regions) // 2]
    kern_ms_region = sorted(r[2] for r in regions)[len(regions) // 2]

    # per-launch event pairs (outside the timed regions): isolates the kernel from inter-launch gaps
    pairs = []
    for i in range(40):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); step(i); b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    bank.check()                                                            # device-side sizing assertions (d_err)
    per_launch = sorted(a.elapsed_time(b) for a, b in pairs)
    kern_ms_pair = per_launch[len(per_launch) // 2]

    lens = bank.last_out_len()
    samples_per_step = nch * (BLOCK // 2)
    alg_bytes = nch * BLOCK + 2 * int(lens.sum())                 # u8 in once + s16 out (2.0267 B/sample)
    achieved = alg_bytes / (kern_ms_region * 1e-3) / 1e9
    value = world * samples_per_step * args.steps / elapsed / 1e6
    per_gpu = [samples_per_step * args.steps / own_med / 1e6]
    if dist is not None:
        mine = torch.tensor([per_gpu[0]], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_gpu = [float(t.item()) for t in every]

    # HBM bytes per launch: NOT measured by this run -- taken from the committed PMC passes of this same command
    # (scripts/gpu_pmc.sh -> profiles/r03_pmc_summary.json; FETCH_SIZE x2 on gfx950 as MI355X_MICROARCH.md
    # prescribes, + WRITE_SIZE), and only when that summary was taken on these very kernel sources.
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, PMC_SUMMARY)) as f:
            pmc = json.load(f)
        same_code = pmc.get("kernel_source_sha16") == kernel_source_hash()
        same_work = abs(pmc["hbm_traffic"]["algorithmic_bytes_per_launch"] - alg_bytes) < 1e-3 * alg_bytes
        if nch == CHANNELS and same_code and same_work:
            traffic = int(pmc["hbm_traffic"]["bytes_per_launch"])
            traffic_src = "%s (committed rocprofv3 --pmc passes of this command on these kernel sources, sha16 %s; not re-measured by this run)" % (
                PMC_SUMMARY, pmc["kernel_source_sha16"])
        elif not same_code:
            traffic_src = "null: %s was taken on other kernel sources" % PMC_SUMMARY
    except (OSError, KeyError, ValueError):
        pass

    # ---- every rank, outside the timed regions: parity bit and clock / power of its own GPU ------------------------------
    def gather_rows(vals):
        """one row of floats per rank -> list of rows (rank order)"""
        if dist is None:
            return [list(vals)]
        mine = torch.tensor(list(vals), dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        return [[float(x) for x in t.tolist()] for t in every]

    launched = bank.last_kernel()
    par = None
    if not args.no_cpu:
        try:
            par = parity_bit(fmd, torch, cfg, bufs, dev_index, stream)
        except Exception as e:
            par = {"ok": False, "error": repr(e), "channels_checked": 0}
        rows = gather_rows([1.0 if par.get("ok") else 0.0, float(par.get("channels_checked", 0))])
        par = dict(par, ok=all(r[0] == 1.0 for r in rows), ranks_checked=len(rows), ranks_ok=[r[0] == 1.0 for r in rows],
                   channels_checked=int(sum(r[1] for r in rows)),
                   seeding="rank r checks channels of its own shard [r * %d, (r + 1) * %d): global channel c is seeded with base + c" % (nch, nch))
    pc = None
    if not args.no_extra:
        fence()                                              # all GPUs of the node load up together: that is the point
        try:
            pc = extra_power_clock(torch, dev_index, step)
        except Exception as e:
            pc = {"error": repr(e)}
        med = lambda d, k: float((d.get(k) or {}).get("median") or float("nan"))
        rows = gather_rows([med(pc, "sclk_mhz"), med(pc, "power_w"), float(pc.get("ms_per_call_during_sampling") or float("nan")),
                            float(pc.get("power_cap_w") or float("nan"))])
        nn = lambda v, nd: None if v != v else round(v, nd)
        pc["per_gpu"] = [{"rank": i, "sclk_mhz": nn(r[0], 0), "power_w": nn(r[1], 0), "ms_per_call": nn(r[2], 4), "power_cap_w": nn(r[3], 0)}
                         for i, r in enumerate(rows)]
        fence()

    if rank == 0:
        res = {
            "metric": "IQ Msamples/s demodulated", "value": round(value, 1), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: %d FM channels x %d B/call @ 2.4 Msps per GPU "
                                   "(downsample %d, %d Hz -> %d Hz), inputs resident in HBM, %d rotating batches"
                                   % (nch, BLOCK, D, FAST, SLOW, args.nbuf),
                       "channels_per_gpu": nch, "block_bytes": BLOCK, "downsample": D, "rate_out": FAST,
                       "rate_resample": SLOW, "audio_per_call": int(lens[0]), "tiling": bank.tiling(),
                       "settle_steps_untimed": args.settle,
                       "runtime": {"hip": getattr(torch.version, "hip", None), "torch": torch.__version__,
                                   "device": torch.cuda.get_device_name(dev_index)},
                       "parallelism": "channels sharded x%d, no collective" % world,
                       "torch_distributed": None if dist is None else {"backend": args.backend, "world_size": world,
                                                                      "collectives": "barrier per fence, all_reduce(MAX) per region, all_gather of the per-GPU rates"}},
            "timing": {"regions": len(regions), "steps_per_region": args.steps, "statistic": "median region, max over ranks",
                       "ms_per_step_min": round(walls[0] / args.steps * 1e3, 4),
                       "ms_per_step_max": round(walls[-1] / args.steps * 1e3, 4),
                       "timed_ms_total": round(sum(walls) * 1e3, 2)},
            "per_gpu_msamples_per_s": [round(v, 1) for v in per_gpu],
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_measured_in_this_run": False,
                         "kernel": launched, "kernel_expected": KERNEL_EXPECTED, "kernel_is_expected": launched == KERNEL_EXPECTED,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         # BASELINE's "% HBM-read roofline" (SURVEY 8d): 2 B per IQ sample only, writes not counted
                         "hbm_read_frac": round(nch * BLOCK / (kern_ms_region * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "kernel_ms_events_region": round(kern_ms_region, 4),
                         "kernel_ms_events_per_launch_median": round(kern_ms_pair, 4),
                         "kernel_ms_events_per_launch_min_max": [round(per_launch[0], 4), round(per_launch[-1], 4)]},
        }
        hb = bound_fields(load_bounds(), "%d,%d,%d" % (D, FAST, SLOW), res["roofline"]["frac"])
        for k in ("valu_issue_frac", "salu_issue_frac", "lds_bank_conflict_share", "bound_source"):
            if k in hb:
                res["roofline"][k] = hb[k]
        if launched != KERNEL_EXPECTED:
            sys.stderr.write("bench.py: the library launched %r, expected %r\n" % (launched, KERNEL_EXPECTED))
        if par is not None:
            res["parity"] = par
        if pc is not None:
            res["extra"] = {"power_clock": pc}
        if world == 1 and not args.no_extra and not args.power_only:
            # what this box's HBM does on a plain device-to-device copy of one input batch (SURVEY 8d asks for the
            # measured reference beside the 8 TB/s spec): bytes read + bytes written over the HIP-event time
            try:
                dst = torch.empty_like(bufs[0])
                for _ in range(3):
                    dst.copy_(bufs[0])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    dst.copy_(bufs[0])
                e1.record()
                torch.cuda.synchronize()
                copy_gbs = 2 * bufs[0].numel() * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
                res["roofline"]["box_reference"] = {
                    "d2d_copy_GBps": round(copy_gbs, 1),
                    "what": "hipMemcpyDtoD (torch copy_) of one %d-byte input batch on this box, read + write bytes / HIP-event time; "
                            "the kernel's `achieved` is %.2f x this" % (bufs[0].numel(), achieved / copy_gbs)}
                del dst
                sk = extra_skeleton(fmd, torch, dev, stream, bufs)
                res["roofline"]["box_reference"]["skeleton"] = sk
                if isinstance(sk.get("headline"), dict) and sk["headline"].get("ms_per_call"):
                    res["roofline"]["box_reference"]["skeleton_ms_headline"] = sk["headline"]["ms_per_call"]
                    res["roofline"]["frac_of_skeleton"] = round(sk["headline"]["ms_per_call"] / kern_ms_region, 4)
                if isinstance(sk.get("cfg_ref"), dict) and sk["cfg_ref"].get("ms_per_call"):
                    res["roofline"]["box_reference"]["skeleton_ms_cfg_ref"] = sk["cfg_ref"]["ms_per_call"]
            except Exception as e:
                res["roofline"]["box_reference"] = {"error": repr(e)}
            bounds = load_bounds()
            side = [("cfg_ref", lambda: extra_cfg_ref(fmd, torch, dev, stream, bufs, bounds)),
                    ("check_per_step", lambda: extra_check_per_step(fmd, torch, bank, bufs, out, cap, stream)),
                    ("check_pipelined", lambda: extra_check_pipelined(fmd, torch, bank, bufs, out, cap, stream)),
                    ("config2_1channel", lambda: extra_config2(fmd, torch, dev, stream)),
                    ("config4_fir", lambda: extra_config4(fmd, torch, dev, stream, False, bounds)),
                    ("config4_fir_demod_fused", lambda: extra_config4(fmd, torch, dev, stream, True, bounds)),
                    ("sink_pcie", lambda: extra_sink_pcie(fmd, dev_index)),
                    ("domain", lambda: extra_domain(fmd, torch, dev, stream, bufs, bounds))]
            for name, fn in side:
                try:
                    res["extra"][name] = fn()
                except Exception as e:          # side lines never break the headline
                    res["extra"][name] = {"error": repr(e)}
            try:                                 # the reference's own rates against THEIR staging skeleton on this box
                sk_ref = res["roofline"]["box_reference"].get("skeleton_ms_cfg_ref")
                if sk_ref and res["extra"]["cfg_ref"].get("ms_per_call"):
                    res["extra"]["cfg_ref"]["frac_of_skeleton"] = round(sk_ref / res["extra"]["cfg_ref"]["ms_per_call"], 4)
            except (KeyError, TypeError, AttributeError):
                pass
        if not args.no_cpu:                      # rank 0 only, after the last timed region; the other ranks wait at the final barrier
            try:
                res["cpu_baseline"] = cpu_baseline(fmd, torch, cfg, bufs[0], target_s=args.cpu_seconds)
            except Exception as e:              # the baseline is reported, never required for the GPU number
                res["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
