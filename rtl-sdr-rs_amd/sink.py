"""Pipelined multi-GPU sink for read_sync buffers (include/fmd.h, fmd_sink_*): new surface over the reference's
receive -> mpsc -> process -> output hand-off (examples/simple_fm.rs:55-60,114-127,150-156)."""
import ctypes as C

import numpy as np

from ._ffi import SINK_CALLBACK, check, lib


class Sink:
    """on_audio(seq, audio_list, status): audio_list[c] is channel c's int16 array of that buffer (copied)."""

    def __init__(self, config, n_channels, nbytes, device_ids=(0,), depth=3, on_audio=None):
        self.n_channels, self.nbytes = int(n_channels), int(nbytes)
        self.on_audio = on_audio
        self.results = []                                   # (seq, [arrays], status) when no callback is given
        self._pending_exc = None                            # first exception raised by on_audio inside the C callback

        def _cb(user, seq, audio, out_len, out_cap, status):
            # ctypes prints and swallows an exception that escapes a callback: a failing consumer would not stop
            # push / pump / drain.  Keep the first one and re-raise it from the call that delivered the buffer.
            try:
                lens = [out_len[c] for c in range(self.n_channels)]
                flat = np.ctypeslib.as_array(audio, shape=(self.n_channels, out_cap))
                rows = [flat[c, :lens[c]].copy() for c in range(self.n_channels)]
                if self.on_audio:
                    self.on_audio(seq, rows, status)
                else:
                    self.results.append((seq, rows, status))
            except BaseException as e:                      # noqa: BLE001 -- re-raised by _reraise()
                if self._pending_exc is None:
                    self._pending_exc = e

        self._cb = SINK_CALLBACK(_cb)                       # keep alive
        ids = (C.c_int32 * len(device_ids))(*device_ids)
        self._h = C.c_void_p()
        check(lib().fmd_sink_new(C.byref(config), self.n_channels, ids, len(device_ids), self.nbytes, depth,
                                 C.cast(self._cb, C.c_void_p), None, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().fmd_sink_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:                                                 # (at interpreter shutdown the module globals may be gone already)
            self.close()
        except Exception:
            pass

    def _reraise(self):
        if self._pending_exc is not None:
            e, self._pending_exc = self._pending_exc, None
            raise e

    def acquire(self):
        """The next slot as a writable uint8 array [n_channels, nbytes] (page-locked memory owned by the sink)."""
        p = C.c_void_p()
        rc = lib().fmd_sink_acquire(self._h, C.byref(p))
        if rc == 0 and self._pending_exc is not None:       # a completion delivered inside acquire failed in on_audio
            lib().fmd_sink_release(self._h)
        self._reraise()
        check(rc)
        raw = (C.c_uint8 * (self.n_channels * self.nbytes)).from_address(p.value)
        return np.frombuffer(raw, dtype=np.uint8).reshape(self.n_channels, self.nbytes)

    def submit(self):
        check(lib().fmd_sink_submit(self._h))

    def release(self):
        """Give the acquired slot back without submitting it (fmd_sink_release): the sink stays usable."""
        check(lib().fmd_sink_release(self._h))

    def push(self, iq):
        """acquire + copy + submit: what receive() does with a freshly read buffer (simple_fm.rs:114-127)."""
        self.acquire()[:] = iq
        self.submit()

    def poll(self):
        n = lib().fmd_sink_poll(self._h)
        self._reraise()
        if n < 0:
            check(n)
        return n

    def drain(self):
        rc = lib().fmd_sink_drain(self._h)
        self._reraise()
        check(rc)

    def info(self):
        cap, nd, fl = C.c_size_t(), C.c_uint32(), C.c_uint32()
        check(lib().fmd_sink_info(self._h, C.byref(cap), C.byref(nd), C.byref(fl)))
        return {"out_cap": cap.value, "n_devices": nd.value, "in_flight": fl.value}

    def f64_stats(self):
        g, p = C.c_uint64(), C.c_uint64()
        check(lib().fmd_sink_f64_stats(self._h, C.byref(g), C.byref(p)))
        return {"guarded": g.value, "patched": p.value}


def pump(sources, sink, max_buffers=None):
    """receive() of the example (simple_fm.rs:100-132) for many streams: every source fills its row of the next slot
    with `read_sync` (src/lib.rs:153; e.g. rtl_tcp_source.RtlTcpSource, any object with read_sync(buf) -> bytes
    written), the slot is submitted, repeat.  A short read on any source ends the run like the reference's
    "samples lost" exit (:122-125).  Returns the number of buffers submitted; completion callbacks run on this
    thread, the rest of them inside the final drain()."""
    if len(sources) != sink.n_channels:
        raise ValueError("one source per channel")
    n = 0
    while max_buffers is None or n < max_buffers:
        slot = sink.acquire()
        if any(src.read_sync(slot[c]) < sink.nbytes for c, src in enumerate(sources)):
            sink.release()                                  # the sink stays usable after a short read
            break
        sink.submit()
        n += 1
    sink.drain()                                            # everything submitted is delivered
    return n
