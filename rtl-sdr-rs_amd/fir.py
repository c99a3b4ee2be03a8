"""Host-side wrapper of the generalised tapped decimating FIR (include/fmd.h, SURVEY 8a row G').

Not a reference interface (the reference's only tapped FIR runs inside the RTL2832U, src/rtlsdr.rs:525-558):
y[m] = sum_t taps[t] * x[decim*m + t] over the rotated + centred IQ stream of each channel.
"""
import ctypes as C

import numpy as np

from ._ffi import DemodState, DeviceConfig, check, lib


class FirBank:
    def __init__(self, taps, decim, n_channels=1, device_id=-1):
        self.taps = np.ascontiguousarray(taps, dtype=np.int16)
        self.decim, self.n_channels = int(decim), int(n_channels)
        self._h = C.c_void_p()
        dev = DeviceConfig(self.n_channels, device_id, 0)
        check(lib().fmd_fir_new(self.taps.ctypes.data_as(C.POINTER(C.c_int16)), self.taps.size, self.decim,
                                C.byref(dev), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().fmd_fir_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:                                                 # (at interpreter shutdown the module globals may be gone already)
            self.close()
        except Exception:
            pass

    def reset(self):
        check(lib().fmd_fir_reset(self._h))

    def out_cap(self, nbytes):
        return int(lib().fmd_fir_out_cap(self.taps.size, self.decim, nbytes))

    def tap_digits(self):
        """1 / 2: the matrix-core form with one (every |tap| <= 127) or two i8 digits per tap; 0: the vector-pipe kernel."""
        return int(lib().fmd_fir_tap_digits(self._h))

    def filter_batch(self, iq):
        """iq uint8 [n_channels, nbytes] -> int32 array [n_channels, n_out, 2] (re, im)."""
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        if iq.ndim != 2 or iq.shape[0] != self.n_channels:
            raise ValueError("iq must be [n_channels, nbytes]")
        cap = max(1, self.out_cap(iq.shape[1]))
        out = np.empty((self.n_channels, cap, 2), dtype=np.int32)
        lens = (C.c_size_t * self.n_channels)()
        check(lib().fmd_fir_filter_batch(self._h, iq.ctypes.data, iq.shape[1], out.ctypes.data, cap, lens))
        return out[:, :lens[0], :].copy()

    def kernel_name(self):
        """The kernel this bank launches, as rocprofv3 --kernel-trace prints it."""
        buf = C.create_string_buffer(128)
        check(lib().fmd_fir_kernel_name(self._h, buf, len(buf)))
        return buf.value.decode()

    def filter_device(self, d_iq, nbytes, d_out, out_cap, stream=None):
        """Enqueue on device pointers.  `stream` must stay alive until the handle's next `filter_device` call has returned
        (stream lifetime rule of include/fmd.h)."""
        n = C.c_size_t(0)
        check(lib().fmd_fir_filter_device(self._h, d_iq, nbytes, d_out, out_cap, C.byref(n), stream))
        return n.value


def auto_shift(taps, limit=2048):
    """Smallest normalisation shift with (128 * sum|taps|) >> shift <= limit.  The default, 2048, is the range of the
    reference chain itself at downsample 16 (|lp| <= 128 * 16): there the fused kernel's f32 discriminator is exact.
    limit=16384 is the most the discriminator admits (three more bits of the filter output, integer form, a few
    percent slower); the library enforces only that bound."""
    g = 128 * int(np.abs(np.asarray(taps, dtype=np.int64)).sum())
    s = 0
    while (g >> s) > limit:
        s += 1
    return s


class FirDemodBank:
    """Tapped FIR -> discriminator -> resampler in one kernel (include/fmd.h, fmd_firdemod_*): Demod::demodulate
    (simple_fm.rs:256-269) with the boxcar replaced by `taps` (decimate by `decim`, normalise by >> shift)."""

    def __init__(self, taps, decim, rate_out, rate_resample, n_channels=1, shift=None, device_id=-1):
        self.taps = np.ascontiguousarray(taps, dtype=np.int16)
        self.decim, self.n_channels = int(decim), int(n_channels)
        self.rate_out, self.rate_resample = int(rate_out), int(rate_resample)
        self.shift = auto_shift(self.taps) if shift is None else int(shift)
        self._h = C.c_void_p()
        dev = DeviceConfig(self.n_channels, device_id, 0)
        check(lib().fmd_firdemod_new(self.taps.ctypes.data_as(C.POINTER(C.c_int16)), self.taps.size, self.decim, self.shift,
                                     self.rate_out, self.rate_resample, C.byref(dev), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().fmd_firdemod_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:                                                 # (at interpreter shutdown the module globals may be gone already)
            self.close()
        except Exception:
            pass

    def reset(self):
        check(lib().fmd_firdemod_reset(self._h))

    def out_cap(self, nbytes):
        return int(lib().fmd_firdemod_out_cap(self.decim, self.rate_out, self.rate_resample, nbytes))

    def demodulate_batch(self, iq):
        """iq uint8 [n_channels, nbytes] -> int16 array [n_channels, n_audio]."""
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        if iq.ndim != 2 or iq.shape[0] != self.n_channels:
            raise ValueError("iq must be [n_channels, nbytes]")
        cap = max(1, self.out_cap(iq.shape[1]))
        out = np.empty((self.n_channels, cap), dtype=np.int16)
        lens = (C.c_size_t * self.n_channels)()
        check(lib().fmd_firdemod_demodulate_batch(self._h, iq.ctypes.data, iq.shape[1], out.ctypes.data, cap, lens))
        return out[:, :lens[0]].copy()

    def demodulate_device(self, d_iq, nbytes, d_out, out_cap, stream=None):
        """Enqueue on device pointers.  `stream` must stay alive until the handle's next `demodulate_device` call or `check` has
        returned (stream lifetime rule of include/fmd.h)."""
        n = C.c_size_t(0)
        check(lib().fmd_firdemod_demodulate_device(self._h, d_iq, nbytes, d_out, out_cap, C.byref(n), stream))
        return n.value

    def check(self):
        check(lib().fmd_firdemod_check(self._h))

    def get_state(self, channel=0):
        s = DemodState()
        check(lib().fmd_firdemod_get_state(self._h, channel, C.byref(s)))
        return s

    def checkpoint(self):
        """The whole bank's carried state (position, resampler phase, per-channel accumulator / last output / filter history)
        as bytes; `resume` on a bank with the same taps, decimation, shift, rates and channel count continues bit for bit."""
        n = lib().fmd_firdemod_checkpoint_size(self._h)
        buf = C.create_string_buffer(n)
        check(lib().fmd_firdemod_checkpoint(self._h, buf, n))
        return buf.raw

    def resume(self, blob):
        blob = bytes(blob)
        check(lib().fmd_firdemod_resume(self._h, blob, len(blob)))

    def f64_stats(self):
        g, p = C.c_uint64(), C.c_uint64()
        check(lib().fmd_firdemod_f64_stats(self._h, C.byref(g), C.byref(p)))
        return {"guarded": g.value, "patched": p.value}

    def tiling(self):
        a, b = C.c_uint32(), C.c_uint32()
        check(lib().fmd_firdemod_tiling(self._h, C.byref(a), C.byref(b)))
        return {"audio_per_tile": a.value, "lds_bytes": b.value}

    def kernel_name(self):
        """The kernel this bank launches, as rocprofv3 --kernel-trace prints it."""
        buf = C.create_string_buffer(128)
        check(lib().fmd_firdemod_kernel_name(self._h, buf, len(buf)))
        return buf.value.decode()
