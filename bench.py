#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s demodulated by the fused HIP path, with its HBM roofline and a CPU baseline.

One "step" = one call of fmd_demod_demodulate_device over one batch of synthetic IQ already
resident in HBM: BASELINE.json configs[2], 4096 FM channels x 262144 B (DEFAULT_BUF_LENGTH,
src/lib.rs:25) at the 2.4 Msps configuration (downsample 10, 240 kHz -> 32 kHz), per GPU.
Channels are independent, so N GPUs = N x 4096 channels with no data-path collective (weak scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHANNELS = 4096
BLOCK = 16 * 16384
D, FAST, SLOW = 10, 240000, 32000
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
KERNEL = "fmd_demod_tile_kernel<5, 256>"   # the dominant kernel of this workload (rocprofv3 --kernel-trace name)


def cpu_baseline(fmd, torch, cfg, iq_dev, target_s=12.0):
    """The oracle (oracle/fm_oracle.c, a C restatement of the reference passes: kind 'port') timed on this
    box's host cores over a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    o = oracle_lib.load()
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
    while total < target_s and passes < 4000:                 # ~12 s of CPU work, bounded
        t = run()
        if t <= 0:
            return None
        total += t
        passes += 1
    samples = passes * chans * calls * (BLOCK // 2)
    # one thread, one channel, same blocks -- about 1 s
    one_t, one_n = 0.0, 0
    while one_t < 1.0 and one_n < 200:
        t = o.lib.fmo_bench_batch(C.byref(ocfg), data.ctypes.data_as(u8p), 1, calls, BLOCK, 1, C.byref(chk), None)
        if t <= 0:
            break
        one_t += t
        one_n += 1
    single = round(one_n * calls * (BLOCK // 2) / one_t / 1e6, 2) if one_t > 0 else None
    return {"value": round(samples / total / 1e6, 2), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "single_thread_value": single,
            "sample": "%d passes x %d channels x %d calls x %d B of the same synthetic workload, %d threads, "
                      "%.1f s of oracle time" % (passes, chans, calls, BLOCK, cores, total)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--channels", type=int, default=CHANNELS)
    ap.add_argument("--nbuf", type=int, default=3, help="distinct input batches rotated through (defeats the 256 MiB L3)")
    ap.add_argument("--kt", type=int, default=0, help="tiling override (audio samples per tile)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / max-over-ranks timing (nccl = RCCL)")
    ap.add_argument("--settle", type=int, default=150,
                    help="untimed steps before the W warm-up steps: the first ~10 ms after an idle period run at "
                         "ramping clocks (measured 0.21-0.23 ms/step vs 0.193 settled); reported in config")
    args = ap.parse_args()

    import torch
    import rtl_sdr_rs_amd as fmd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # one rank per GPU; `--backend gloo` (ranks may then share a GPU) exists to exercise this launch path on a
    # single-GPU box -- the driver's multi-GPU runs use the default, RCCL ("nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    if fmd.device_count() < 1:
        raise RuntimeError("bench.py: no gfx950 device; the product has no CPU path")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    local_rank = dev_index

    cfg = fmd.DemodConfig(FAST, FAST, SLOW, D, max(1, (1 << 15) // (128 * D)))
    nch = args.channels
    lo, hi = fmd.shard.channel_range(nch * world, world, rank)          # this rank's global channel ids
    bank = fmd.DemodBank(cfg, nch, device_id=local_rank)
    if args.kt:
        bank.set_tiling(args.kt)
    cap = bank.out_cap(BLOCK)
    stream = torch.cuda.current_stream().cuda_stream
    bufs = []
    for b in range(args.nbuf):
        t = torch.empty((nch, BLOCK), dtype=torch.uint8, device=dev)
        fmd.synth.fill_device(t.data_ptr(), nch, BLOCK, sample_offset=b * (BLOCK // 2), device_id=local_rank,
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

    for i in range(args.settle):
        step(i)
    for i in range(args.warmup):
        step(i)
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step(args.warmup + i)
    ev1.record()
    fence()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kern_ms_region = ev0.elapsed_time(ev1) / args.steps            # HIP events on the launch stream

    # per-launch event pairs (outside the timed region): isolates the kernel from inter-launch gaps
    pairs = []
    for i in range(min(args.steps, 20)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); step(i); b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    per_launch = sorted(a.elapsed_time(b) for a, b in pairs)
    kern_ms_pair = per_launch[len(per_launch) // 2]

    lens = bank.last_out_len()
    samples_per_step = nch * (BLOCK // 2)
    alg_bytes = nch * BLOCK + 2 * int(lens.sum())                 # u8 in once + s16 out (2.0267 B/sample)
    achieved = alg_bytes / (kern_ms_region * 1e-3) / 1e9
    value = world * samples_per_step * args.steps / elapsed / 1e6

    # HBM bytes per launch from the committed PMC passes of this same command (scripts/gpu_pmc.sh ->
    # profiles/r01_pmc_summary.json; FETCH_SIZE x2 on gfx950 as MI355X_MICROARCH.md prescribes, + WRITE_SIZE).
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")) as f:
            pmc = json.load(f)
        if nch == CHANNELS and abs(pmc["hbm_traffic"]["algorithmic_bytes_per_launch"] - alg_bytes) < 1e-3 * alg_bytes:
            traffic = int(pmc["hbm_traffic"]["bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        pass

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
                       "parallelism": "channels sharded x%d, no collective" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": KERNEL,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         # BASELINE's "% HBM-read roofline" (SURVEY 8d): 2 B per IQ sample only, writes not counted
                         "hbm_read_frac": round(nch * BLOCK / (kern_ms_region * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "kernel_ms_events_region": round(kern_ms_region, 4),
                         "kernel_ms_events_per_launch_median": round(kern_ms_pair, 4)},
        }
        if world == 1 and not args.no_cpu:
            try:
                res["cpu_baseline"] = cpu_baseline(fmd, torch, cfg, bufs[0])
            except Exception as e:              # the baseline is reported, never required for the GPU number
                res["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
