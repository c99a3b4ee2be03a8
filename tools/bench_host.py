#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (fmd_demod_demodulate_batch: H2D + kernel + D2H per
call), with pageable numpy buffers and with page-locked buffers from fmd_host_alloc.  Reported in DESIGN.md
only -- never bench.py's `value`."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtl_sdr_rs_amd as fmd

nch, N = 1024, fmd.DEFAULT_BUF_LENGTH
cfg = fmd.DemodConfig(240000, 240000, 32000, 10, 25)
bank = fmd.DemodBank(cfg, nch)
cap = bank.out_cap(N)
src = np.ascontiguousarray(np.tile(fmd.synth.synth_iq(8, N), (nch // 8, 1)))
res = {"workload": "%d channels x %d B per call, host buffers: H2D + kernel + D2H" % (nch, N)}


def run(iq, out, steps=10):
    for _ in range(3):
        bank.demodulate_batch_into(iq, out)
    t0 = time.perf_counter()
    for _ in range(steps):
        bank.demodulate_batch_into(iq, out)
    dt = (time.perf_counter() - t0) / steps
    return {"ms_per_call": round(dt * 1e3, 3), "iq_msamples_per_s": round(nch * N / 2 / dt / 1e6, 1),
            "host_GBps": round(nch * N / dt / 1e9, 2)}


res["pageable"] = run(src, np.empty((nch, cap), np.int16))
pin_in, pin_out = fmd.PinnedBuffer((nch, N), np.uint8), fmd.PinnedBuffer((nch, cap), np.int16)
pin_in.array[:] = src
res["pinned"] = run(pin_in.array, pin_out.array)
print(json.dumps(res))

# the pipelined sink (fmd_sink_*): the same buffers through a ring of 3 page-locked slots; H2D / kernel / D2H overlap.
# Submission only touches the GPU queue, so the per-buffer time the CALLER sees is the link time alone.
done = []
sink = fmd.Sink(cfg, nch, N, device_ids=[0], depth=3, on_audio=lambda seq, rows, status: None)
import ctypes as C
from rtl_sdr_rs_amd._ffi import lib, check
raw_cb = sink._cb                                   # replace the copying callback by a counting one for timing
count = [0]
def _count(user, seq, audio, out_len, out_cap, status):
    count[0] += 1
from rtl_sdr_rs_amd._ffi import SINK_CALLBACK
sink.close()
cb = SINK_CALLBACK(_count)
h = C.c_void_p()
ids = (C.c_int32 * 1)(0)
check(lib().fmd_sink_new(C.byref(cfg), nch, ids, 1, N, 3, C.cast(cb, C.c_void_p), None, C.byref(h)))
def push():
    p = C.c_void_p()
    check(lib().fmd_sink_acquire(h, C.byref(p)))
    C.memmove(p.value, src.ctypes.data, nch * N)       # stands for read_sync writing into the slot
    check(lib().fmd_sink_submit(h))
for _ in range(4):
    push()
check(lib().fmd_sink_drain(h))
steps = 20
t0 = time.perf_counter()
for _ in range(steps):
    push()
check(lib().fmd_sink_drain(h))
dt = (time.perf_counter() - t0) / steps
res["sink_depth3_incl_memcpy_into_slot"] = {"ms_per_buffer": round(dt * 1e3, 3), "iq_msamples_per_s": round(nch * N / 2 / dt / 1e6, 1),
                                            "host_GBps": round(nch * N / dt / 1e9, 2), "delivered": count[0]}
# the same without refilling the slots (acquire + submit only): what the pipeline itself sustains
def push_nofill():
    p = C.c_void_p()
    check(lib().fmd_sink_acquire(h, C.byref(p)))
    check(lib().fmd_sink_submit(h))
t0 = time.perf_counter()
for _ in range(steps):
    push_nofill()
check(lib().fmd_sink_drain(h))
dt = (time.perf_counter() - t0) / steps
res["sink_depth3_slots_prefilled"] = {"ms_per_buffer": round(dt * 1e3, 3), "host_GBps": round(nch * N / dt / 1e9, 2)}
tmp = np.empty_like(src)
t0 = time.perf_counter()
for _ in range(5):
    C.memmove(tmp.ctypes.data, src.ctypes.data, nch * N)
res["host_memmove_ms_per_buffer"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
lib().fmd_sink_free(h)
print(json.dumps(res))
