"""Host-side mirror of the reference's Demod interface (examples/simple_fm.rs:172-269)."""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import DemodConfig, DemodState, DeviceConfig, RadioConfig, check, lib


def device_count():
    n = C.c_int(0)
    rc = lib().fmd_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def optimal_settings(freq, rate, rate_resample=32000):
    """optimal_settings(freq, rate) -> (RadioConfig, DemodConfig), simple_fm.rs:189-214."""
    r, d = RadioConfig(), DemodConfig()
    check(lib().fmd_optimal_settings(freq, rate, rate_resample, C.byref(r), C.byref(d)))
    return r, d


def out_cap(config, nbytes):
    return int(lib().fmd_out_cap(C.byref(config), nbytes))


class PinnedBuffer:
    """Page-locked host memory from fmd_host_alloc, viewed as a numpy array (`.array`): read buffers taken from
    here let the host entry points DMA without the runtime's staging copy."""

    def __init__(self, shape, dtype=np.uint8):
        self.shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self._p = C.c_void_p()
        check(lib().fmd_host_alloc(self.nbytes, C.byref(self._p)))
        raw = (C.c_uint8 * self.nbytes).from_address(self._p.value)
        self.array = np.frombuffer(raw, dtype=self.dtype).reshape(self.shape)

    def close(self):
        if getattr(self, "_p", None) is not None and self._p:
            self.array = None
            check(lib().fmd_host_free(self._p))
            self._p = C.c_void_p()

    def __del__(self):
        try:                                                 # (at interpreter shutdown the module globals may be gone already)
            self.close()
        except Exception:
            pass


class DemodBank:
    """n_channels independent reference `Demod`s living on one MI355X."""

    def __init__(self, config, n_channels=1, device_id=-1):
        self.config = config
        self.n_channels = int(n_channels)
        self._h = C.c_void_p()
        dev = DeviceConfig(self.n_channels, device_id, 0)
        check(lib().fmd_demod_new(C.byref(config), C.byref(dev), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().fmd_demod_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:                                                 # (at interpreter shutdown the module globals may be gone already)
            self.close()
        except Exception:
            pass

    def reset(self):
        check(lib().fmd_demod_reset(self._h))

    def out_cap(self, nbytes):
        return out_cap(self.config, nbytes)

    def demodulate_batch(self, iq):
        """iq: uint8 array [n_channels, nbytes] on the host -> list of int16 arrays."""
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        if iq.ndim != 2 or iq.shape[0] != self.n_channels:
            raise ValueError("iq must be [n_channels, nbytes]")
        nbytes = iq.shape[1]
        cap = max(1, self.out_cap(nbytes))
        out = np.empty((self.n_channels, cap), dtype=np.int16)
        lens = (C.c_size_t * self.n_channels)()
        check(lib().fmd_demod_demodulate_batch(self._h, iq.ctypes.data, nbytes, out.ctypes.data, cap, lens))
        return [out[c, :lens[c]].copy() for c in range(self.n_channels)]

    def demodulate_batch_into(self, iq, out):
        """No-copy form: iq uint8 [n_channels, nbytes], out int16 [n_channels, cap] (both C-contiguous host arrays,
        e.g. PinnedBuffer.array).  Returns the per-channel sample counts."""
        if iq.dtype != np.uint8 or out.dtype != np.int16 or not iq.flags.c_contiguous or not out.flags.c_contiguous:
            raise ValueError("need C-contiguous uint8 input and int16 output")
        if iq.ndim != 2 or out.ndim != 2 or iq.shape[0] != self.n_channels or out.shape[0] != self.n_channels:
            raise ValueError("iq / out must be [n_channels, *]")
        lens = (C.c_size_t * self.n_channels)()
        check(lib().fmd_demod_demodulate_batch(self._h, iq.ctypes.data, iq.shape[1], out.ctypes.data, out.shape[1], lens))
        return np.array(lens[:], dtype=np.int64)

    def demodulate_device(self, d_iq, nbytes, d_out, out_cap_, d_out_len=None, stream=None):
        """Enqueue on device pointers (ints).  Returns immediately.
        STREAM LIFETIME (include/fmd.h): `stream` (a hipStream_t as an int; None = the default stream) must stay alive until this
        handle's NEXT `demodulate_device` call or completion point (`check`, `check_prev`, `get_state`, `set_state`, a host entry)
        has returned -- the library touches the most recent launch's stream once more there.  A stream taken from a pool that may
        destroy it (torch's stream objects going out of scope) has to be kept referenced that long."""
        check(lib().fmd_demod_demodulate_device(self._h, d_iq, nbytes, d_out, out_cap_, d_out_len, stream))

    def check(self):
        """fmd_demod_check: wait for the handle's launches, surface device assertions, settle guarded f64 samples."""
        check(lib().fmd_demod_check(self._h))

    def check_prev(self):
        """fmd_demod_check_prev = check_behind(1)."""
        check(lib().fmd_demod_check_prev(self._h))

    def check_behind(self, back=2):
        """fmd_demod_check_behind: the completion point `back` launches back (0: `check`; 1, 2) -- waits for launch n - back while the
        newer ones run and settles its f64 samples.  Until a launch has been settled its output buffer stays allocated and unread and
        its input buffer unmodified (include/fmd.h); launches are settled in order; the stream of the handle's most recent launch must
        be alive, as for `check` (or: `set_event_ordering`)."""
        check(lib().fmd_demod_check_behind(self._h, int(back)))

    def set_event_ordering(self, on=True):
        """fmd_demod_set_event_ordering: record an event behind every launch and wait on events only -- the streams handed to
        `demodulate_device` need not outlive the call (+2 - 3 % per launch)."""
        check(lib().fmd_demod_set_event_ordering(self._h, 1 if on else 0))

    def f64_stats(self):
        g, p = C.c_uint64(), C.c_uint64()
        check(lib().fmd_demod_f64_stats(self._h, C.byref(g), C.byref(p)))
        return {"guarded": g.value, "patched": p.value}

    def last_out_len(self):
        lens = (C.c_size_t * self.n_channels)()
        check(lib().fmd_demod_last_out_len(self._h, lens))
        return np.array(lens[:], dtype=np.int64)

    def get_state(self, channel=0):
        s = DemodState()
        check(lib().fmd_demod_get_state(self._h, channel, C.byref(s)))
        return s

    def set_state(self, channel, state):
        check(lib().fmd_demod_set_state(self._h, channel, C.byref(state)))

    def tiling(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib().fmd_demod_tiling(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"audio_per_tile": a.value, "lds_bytes": b.value, "block_threads": c.value}

    def tiling_plan(self):
        """fmd_demod_tiling_plan: 0 the caller's tile, 1 the 20 KB budget, 2 the 15.5 ... 17.3 KB window (LDS kernels)."""
        return int(lib().fmd_demod_tiling_plan(self._h))

    def last_kernel(self):
        """Name of the kernel the most recent launch ran, as rocprofv3 --kernel-trace prints it ('' before the first launch)."""
        buf = C.create_string_buffer(128)
        check(lib().fmd_demod_last_kernel(self._h, buf, len(buf)))
        return buf.value.decode()

    def set_block_len(self, block_bytes):
        """Treat every call as nbytes / block_bytes consecutive reference calls (0 = off): fmd_demod_set_block_len."""
        check(lib().fmd_demod_set_block_len(self._h, block_bytes))

    def set_tiling(self, audio_per_tile):
        check(lib().fmd_demod_set_tiling(self._h, audio_per_tile))


class Demod(DemodBank):
    """struct Demod (simple_fm.rs:232-269): one stream.  demodulate(buf) -> int16 array."""

    def __init__(self, config, device_id=-1):
        super().__init__(config, 1, device_id)

    def demodulate(self, buf):
        buf = np.ascontiguousarray(np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray)) else buf,
                                   dtype=np.uint8).reshape(-1)
        cap = max(1, self.out_cap(buf.size))
        out = np.empty(cap, dtype=np.int16)
        n = C.c_size_t(0)
        check(lib().fmd_demod_demodulate(self._h, buf.ctypes.data, buf.size, out.ctypes.data, cap, C.byref(n)))
        return out[:n.value].copy()
