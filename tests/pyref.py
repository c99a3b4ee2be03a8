"""Independent pure-Python restatement of the simple_fm demodulation chain (test infrastructure).

Written from the semantics of examples/simple_fm.rs:256-426 (ccostes/rtl-sdr-rs v0.3.1) with
Python big integers and explicit two's-complement wrapping, deliberately NOT sharing code
with oracle/fm_oracle.c: two restatements that agree with each other and with the
reference's three KATs are the strongest pin available for the parts of the path the
reference does not test (rotate_90, centring, state carry, i32 wrap, f64 sample).
Small inputs only -- it is a per-sample Python loop.
"""
import math


def wrap32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def wrap16(v):
    v &= 0xFFFF
    return v - (1 << 16) if v & 0x8000 else v


def tdiv(a, b):
    """Rust/C integer division: truncate toward zero."""
    q = abs(a) // abs(b)
    return q if (a < 0) == (b < 0) else -q


def optimal_settings(freq, rate, rate_resample=32000):
    """simple_fm.rs:189-214 (u32 arithmetic)."""
    downsample = 1_000_000 // rate + 1
    capture_rate = (downsample * rate) & 0xFFFFFFFF
    capture_freq = (freq + capture_rate // 4) & 0xFFFFFFFF
    output_scale = max(1, (1 << 15) // (128 * downsample))
    radio = {"capture_freq": capture_freq, "capture_rate": capture_rate}
    demod = {"rate_in": rate, "rate_out": rate, "rate_resample": rate_resample,
             "downsample": downsample, "output_scale": output_scale}
    return radio, demod


def rotate_90(buf):
    """simple_fm.rs:282-298; per 8 bytes -> [b0, b1, 255-b3, b2, 255-b4, 255-b5, b7, 255-b6]."""
    if len(buf) % 8:
        raise IndexError("len % 8 != 0")
    out = bytearray(len(buf))
    for i in range(0, len(buf), 8):
        b = buf[i:i + 8]
        out[i:i + 8] = bytes([b[0], b[1], 255 - b[3], b[2], 255 - b[4], 255 - b[5], b[7], 255 - b[6]])
    return bytes(out)


def fast_atan2(y, x):
    """simple_fm.rs:383-405 incl. the i64->i32 truncation before the divide."""
    pi4, pi34 = 1 << 12, 3 * (1 << 12)
    if x == 0 and y == 0:
        return 0
    yabs = wrap32(-y) if y < 0 else y
    if x >= 0:
        num = wrap32(pi4 * wrap32(x - yabs))
        angle = wrap32(pi4 - tdiv(num, wrap32(x + yabs)))
    else:
        num = wrap32(pi4 * wrap32(x + yabs))
        angle = wrap32(pi34 - tdiv(num, wrap32(yabs - x)))
    return wrap32(-angle) if y < 0 else angle


def mul_conj(a, b):
    (ar, ai), (br, bi) = a, b
    return wrap32(ar * br + ai * bi), wrap32(ai * br - ar * bi)


def polar_discriminant(a, b):
    """simple_fm.rs:370-374."""
    re, im = mul_conj(a, b)
    angle = math.atan2(float(im), float(re))
    return int(angle / math.pi * float(1 << 14))       # int() truncates toward zero


def polar_discriminant_fast(a, b):
    re, im = mul_conj(a, b)
    return fast_atan2(im, re)


class Demod:
    """struct Demod + impl, simple_fm.rs:232-427."""

    def __init__(self, downsample, rate_out, rate_resample):
        self.downsample, self.rate_out, self.rate_resample = downsample, rate_out, rate_resample
        self.prev_index = 0
        self.now_lpr = 0
        self.prev_lpr_index = 0
        self.lp_now = (0, 0)
        self.demod_pre = (0, 0)

    def state(self):
        return {"prev_index": self.prev_index, "now_lpr": self.now_lpr,
                "prev_lpr_index": self.prev_lpr_index,
                "lp_now": list(self.lp_now), "demod_pre": list(self.demod_pre)}

    def low_pass_complex(self, buf):
        res = []
        for s in buf:
            self.lp_now = (wrap32(self.lp_now[0] + s[0]), wrap32(self.lp_now[1] + s[1]))
            self.prev_index += 1
            if self.prev_index < self.downsample:
                continue
            res.append(self.lp_now)
            self.lp_now = (0, 0)
            self.prev_index = 0
        return res

    def fm_demod(self, buf):
        assert len(buf) > 1
        res = [wrap16(polar_discriminant(buf[0], self.demod_pre))]
        for i in range(1, len(buf)):
            res.append(wrap16(polar_discriminant_fast(buf[i], buf[i - 1])))
        self.demod_pre = buf[-1]
        return res

    def low_pass_real(self, buf):
        res = []
        slow, fast = self.rate_resample, self.rate_out
        for v in buf:
            self.now_lpr = wrap32(self.now_lpr + v)
            self.prev_lpr_index += slow
            if self.prev_lpr_index < fast:
                continue
            res.append(wrap16(tdiv(self.now_lpr, fast // slow)))
            self.prev_lpr_index -= fast
            self.now_lpr = 0
        return res

    def demodulate(self, buf):
        rot = rotate_90(bytes(buf))
        signed = [b - 127 for b in rot]
        cplx = [(signed[i], signed[i + 1]) for i in range(0, len(signed) - 1, 2)]
        return self.low_pass_real(self.fm_demod(self.low_pass_complex(cplx)))
