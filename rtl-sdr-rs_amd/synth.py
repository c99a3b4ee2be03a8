"""Deterministic integer-only synthetic FM source (stands in for the reference's absent capture.bin).

numpy mirror of fmd_synth_kernel (csrc/fmd_kernels.hip): the same bytes on CPU and GPU, no libm.
Signal: one FM carrier at -Fs/4 (what the offset tuning of simple_fm.rs:194-195 delivers; rotate_90
shifts it back to DC), sinusoidal modulation (per-channel pitch), uniform noise, offset-binary u8.
"""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)

DEFAULTS = dict(seed=0x05D50001, amplitude=100, noise=8, dev_q32=134217728, mod_period=2400)
# dev_q32 = 75 kHz / 2.4 Msps in Q32 (= 0.03125 * 2^32); mod_period 2400 samples = 1 kHz at 2.4 Msps


def _mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z ^ (z >> np.uint64(30)); z = z * _M1
        z = z ^ (z >> np.uint64(27)); z = z * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _isin_q15(phase_u32):
    xs = (np.asarray(phase_u32, dtype=np.uint32).view(np.int32) >> 16).astype(np.int64)
    ax = np.abs(xs)
    y = (xs * (32768 - ax)) >> 13
    y2 = (y * np.abs(y)) >> 15
    return y + (((y2 - y) * 7373) >> 15)


def synth_iq(n_channels, nbytes, sample_offset=0, seed=DEFAULTS["seed"], amplitude=DEFAULTS["amplitude"],
             noise=DEFAULTS["noise"], dev_q32=DEFAULTS["dev_q32"], mod_period=DEFAULTS["mod_period"],
             first_channel=0):
    """uint8 array [n_channels, nbytes]; channel index c = first_channel + row."""
    assert nbytes % 8 == 0
    ns = nbytes // 2
    out = np.empty((n_channels, nbytes), dtype=np.uint8)
    n = np.arange(ns, dtype=np.uint64) + np.uint64(sample_offset)
    mod_step = (1 << 32) // mod_period
    beta_q16 = (dev_q32 * mod_period * 10430) >> 32
    span = 2 * noise + 1
    with np.errstate(over="ignore"):
        for row in range(n_channels):
            c = first_channel + row
            hc = _mix64(np.uint64((seed + c + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF))
            chan_phase = np.uint32(int(hc) & 0xFFFFFFFF)
            mod_step_c = np.uint64((mod_step + (c % 61) * (mod_step >> 6)) & 0xFFFFFFFF)
            am = ((n * mod_step_c).astype(np.uint32) + np.uint32(int(hc) >> 32)).astype(np.uint32)
            dphi = ((np.int64(beta_q16) * _isin_q15(am) * 2) & 0xFFFFFFFF).astype(np.uint32)
            theta = ((n * np.uint64(0xC0000000)).astype(np.uint32) + chan_phase + dphi).astype(np.uint32)
            I = (amplitude * _isin_q15(theta + np.uint32(0x40000000))) >> 15
            Q = (amplitude * _isin_q15(theta)) >> 15
            hz = _mix64(hc ^ (n * _GOLD))
            nI = (((hz & np.uint64(0xFFFF)) * np.uint64(span)) >> np.uint64(16)).astype(np.int64) - noise
            nQ = ((((hz >> np.uint64(16)) & np.uint64(0xFFFF)) * np.uint64(span)) >> np.uint64(16)).astype(np.int64) - noise
            out[row, 0::2] = np.clip(127 + I + nI, 0, 255).astype(np.uint8)
            out[row, 1::2] = np.clip(127 + Q + nQ, 0, 255).astype(np.uint8)
    return out


def params_struct(seed=DEFAULTS["seed"], amplitude=DEFAULTS["amplitude"], noise=DEFAULTS["noise"],
                  dev_q32=DEFAULTS["dev_q32"], mod_period=DEFAULTS["mod_period"]):
    from ._ffi import SynthParams
    return SynthParams(seed, amplitude, noise, dev_q32, mod_period)


def fill_device(d_iq, n_channels, nbytes, sample_offset=0, device_id=-1, stream=None, **kw):
    """Fill a device buffer [n_channels, nbytes] (pointer as int) with the same signal on the GPU."""
    import ctypes as C
    from ._ffi import check, lib
    p = params_struct(**kw)
    check(lib().fmd_synth_fill_device(device_id, d_iq, n_channels, nbytes, sample_offset, C.byref(p), stream))
