"""The f32 form of polar_discriminant_fast / fast_atan2 (simple_fm.rs:377-405) used by the tile kernel for
downsample <= 16 (disc_f32, rtl-sdr-rs_amd/csrc/fmd_device.h), replayed step by step in numpy float32 --
same operations, same constants, same order -- with the hardware reciprocal (v_rcp_f32: 1 ulp) pushed to BOTH ends
of its error band, against the integer definition (vectorised here, itself checked against the C oracle).
Every input class the kernel can produce: products a * conj(b) of boxcar sums up to |lp| = 128 * 16, the i32 wrap
ties (s = 2^19 mod 2^20), exact quotients, both axes, (0, 0); and the same at 128 * 18, where the f32 form must FAIL
(|s| above 2^23 loses the + 0.5 of the wrap step), which is why the kernel stops at 16."""
import ctypes as C

import numpy as np
import pytest

F = np.float32
LIM = 128 * 16                      # |lp| <= 128 * D, D <= FMD_DISC_F32_MAX_D
XMAX = 2 * LIM * LIM                # |x|, |y| <= 2 (128 D)^2 = 2^23
DENMAX = int(1.41422 * XMAX)        # |x| + |y| <= sqrt(2) |a| |b| < 2^24


def fast_atan2_ref(y, x):
    """Demod::fast_atan2 on int64 arrays with the `as i32` wrap of the product (simple_fm.rs:383-405)."""
    y = y.astype(np.int64); x = x.astype(np.int64)
    yabs = np.abs(y)
    pos = x >= 0
    num = np.where(pos, x - yabs, x + yabs) * 4096
    num = ((num + 2**31) % 2**32) - 2**31                      # (.. as i64 * 4096) as i32
    den = np.where(pos, x + yabs, yabs - x)
    safe = np.where(den == 0, 1, den)
    q = np.sign(num) * (np.abs(num) // safe)                   # Rust `/`: truncation toward zero
    angle = np.where(pos, 4096, 12288) - q
    res = np.where(y < 0, -angle, angle)
    return np.where((x == 0) & (y == 0), 0, res)


def disc_f32_model(x, y, rcp_ulps, wrap1=False):
    """disc_f32 after the two dot products, in float32; rcp = correctly rounded 1/d nudged by rcp_ulps ulps.
    wrap1: the single-point form of the i32 wrap the kernels use at downsample 4 (disc_f32_xy<.., WRAP1>)."""
    xf, yf = x.astype(F), y.astype(F)
    den = np.abs(xf) + np.abs(yf)
    t = np.abs(xf) - np.abs(yf)
    sx = xf.view(np.uint32) & np.uint32(0x80000000)
    s = (t.view(np.uint32) ^ sx).view(F)
    big = F(13194139533312.0)
    sp = s - (((s + F(0.5)) + big) - big)
    if wrap1:
        t1 = np.clip(s - F(524287.0), F(0), F(1))               # v_sub_f32 with the clamp modifier: exact, s is an integer
        sp = (t1.astype(np.float64) * -1048576.0 + s.astype(np.float64)).astype(F)      # one fma: exact product and sum, one rounding (of an integer < 2^24)
    d = den + F(2.0 ** -30)
    rc = (1.0 / d.astype(np.float64)).astype(F)
    for _ in range(abs(rcp_ulps)):
        rc = np.nextafter(rc, F(np.inf) if rcp_ulps > 0 else F(0))
    m = np.abs(sp) * F(4096.0)                                  # exact: a power-of-two scale, <= 2^31
    k = np.rint(m * rc)                                         # v_rndne_f32 of one f32 product
    nz = den > 0
    assert np.all(np.abs(k.astype(np.float64)[nz] - m.astype(np.float64)[nz] / den.astype(np.float64)[nz]) < 0.5 + 4096 * (abs(rcp_ulps) + 2) * 2.0 ** -23)
    d_exact = k.astype(np.float64) * den.astype(np.float64) - m.astype(np.float64)   # integer, |.| <= den < 2^24.5: exact in f64
    d = d_exact.astype(F)                                       # one fma: exact product and sum, one rounding
    assert np.all((d >= F(1)) == (d_exact >= 1)) and np.all((d <= F(0)) == (d_exact <= 0))
    q = k - np.clip(d, F(0), F(1))                              # the fma's clamp modifier
    qs = (q.view(np.uint32) ^ (sp.view(np.uint32) & np.uint32(0x80000000))).view(F)
    base = F(8192.0) - (np.uint32(0x45800000) ^ sx).view(F)
    res = ((base - qs).view(np.uint32) ^ (yf.view(np.uint32) & np.uint32(0x80000000))).view(F)
    out = res * np.clip(den + den, F(0), F(1))
    assert np.all(out == np.trunc(out))
    return out.astype(np.int64)


def products(ar, ai, br, bi):
    return ar * br + ai * bi, ai * br - ar * bi                # c = a * conj(b): (re, im)


def cases(LIM=LIM):
    XMAX = 2 * LIM * LIM
    DENMAX = int(1.41422 * XMAX)
    rng = np.random.default_rng(2026)
    xs, ys = [], []
    n = 400000
    a = rng.integers(-LIM, LIM + 1, (4, n))
    x, y = products(*a); xs.append(x); ys.append(y)
    a = rng.choice([-LIM, -LIM + 1, -1, 0, 1, LIM - 1, LIM], (4, 50000))      # corners: largest magnitudes, axes, zero
    x, y = products(*a); xs.append(x); ys.append(y)
    a = rng.integers(-60, 61, (4, 100000))                                     # weak signals: small den, no wrap
    x, y = products(*a); xs.append(x); ys.append(y)
    # the wrap ties: |x| - |y| = 2^19 (2k + 1) exactly, both signs of x and y
    k = rng.integers(0, 1 + XMAX // 2**20, 60000)
    yy = rng.integers(0, XMAX // 2, 60000)
    xx = yy + 2**19 * (2 * k + 1)
    ok = (xx <= XMAX) & (xx + yy <= DENMAX)
    sgx, sgy = rng.choice([-1, 1], ok.sum()), rng.choice([-1, 1], ok.sum())
    xs.append(xx[ok] * sgx); ys.append(yy[ok] * sgy)
    # exact quotients: den divides 4096 * sp
    den = 2 ** rng.integers(1, 22, 40000)
    sp = (rng.integers(0, 4097, 40000) * den) // 4096
    xx = (den + sp) // 2; yy = den - xx
    xs.append(xx * rng.choice([-1, 1], 40000)); ys.append(yy * rng.choice([-1, 1], 40000))
    # near-exact quotients around every boundary: 4096 * t = q * den + {-1, 0, +1 ...}
    den = rng.integers(1, DENMAX, 200000)
    q = rng.integers(0, 4097, 200000)
    t = (q * den + rng.integers(-3, 4, 200000) + 4095) // 4096
    t = np.clip(t, 0, den)
    xx = (den + t) // 2; yy = den - xx                          # |x| + |y| = den, |x| - |y| ~ t
    xs.append(xx * rng.choice([-1, 1], 200000)); ys.append(yy * rng.choice([-1, 1], 200000))
    xs.append(np.array([0, 0, 0, 5, -5, 1, -1, 2**22, -2**22, 524288, 524287, -524288, 0, 0]))
    ys.append(np.array([0, 7, -7, 0, 0, 1, -1, 0, 0, 0, 0, 0, 524288, -524288]))
    x, y = np.concatenate(xs), np.concatenate(ys)
    keep = (np.abs(x) <= XMAX) & (np.abs(y) <= XMAX) & (np.abs(x) + np.abs(y) <= DENMAX)
    return x[keep], y[keep]


def test_reference_formula_matches_oracle(oracle):
    x, y = cases()
    idx = np.random.default_rng(1).choice(x.size, 20000, replace=False)
    ref = fast_atan2_ref(y[idx], x[idx])
    got = np.array([oracle.lib.fmo_fast_atan2(int(b), int(a)) for a, b in zip(x[idx], y[idx])])
    assert np.array_equal(ref, got)
    assert oracle.lib.fmo_fast_atan2(0, 524288) == 8192 and oracle.lib.fmo_fast_atan2(0, 524287) == 0


@pytest.mark.parametrize("rcp_ulps", [-64, -2, -1, 0, 1, 2, 64])     # v_rcp_f32 is specified to 1 ulp; the nearest-integer form has margin to spare
def test_f32_discriminator_is_exact(rcp_ulps):
    x, y = cases()
    assert x.size > 800000
    ref = fast_atan2_ref(y, x)
    got = disc_f32_model(x, y, rcp_ulps)
    bad = np.nonzero(ref != got)[0]
    assert bad.size == 0, [(int(x[i]), int(y[i]), int(ref[i]), int(got[i])) for i in bad[:8]]
    # the wrap is really exercised, and so is the +1 correction
    assert np.count_nonzero(np.abs(np.abs(x) - np.abs(y)) >= 2**19) > 100000


@pytest.mark.parametrize("rcp_ulps", [-2, 0, 2])
def test_single_point_wrap_at_downsample_4(rcp_ulps):
    """disc_f32_xy<.., WRAP1>: with |lp| <= 512 the product of `(4096 * s) as i32` wraps for s = +2^19 alone (a = b = (512, 512)); the
    kernels' two-instruction form is checked on every input class of `cases` at that limit plus the whole neighbourhood of the wrap point."""
    x, y = cases(512)
    ext = np.arange(2**19 - 40, 2**19 + 1)
    x = np.concatenate([x, ext, -ext, ext - 7, -(ext - 7)]); y = np.concatenate([y, np.zeros(2 * ext.size, np.int64), np.full(ext.size, 7), np.full(ext.size, -7)])
    s = np.where(x >= 0, x - np.abs(y), x + np.abs(y))
    assert s.max() == 2**19 and s.min() == -2**19 and np.count_nonzero(s == 2**19) >= 1      # the domain: [-2^19, 2^19]
    ref = fast_atan2_ref(y, x)
    got = disc_f32_model(x, y, rcp_ulps, wrap1=True)
    bad = np.nonzero(ref != got)[0]
    assert bad.size == 0, [(int(x[i]), int(y[i]), int(ref[i]), int(got[i])) for i in bad[:8]]
    assert fast_atan2_ref(np.array([0]), np.array([2**19]))[0] == 8192                         # the reference's own value at the wrap point (:397)
    # one step outside the domain (downsample 5) the single-point form must fail: it is tied to the factor
    x5, y5 = cases(640)
    assert np.count_nonzero(disc_f32_model(x5, y5, 0, wrap1=True) != fast_atan2_ref(y5, x5)) > 0


def test_f32_discriminator_limit_is_where_the_kernel_stops():
    """One step beyond FMD_DISC_F32_MAX_D = 16 (boxcar sums up to 128 * 18) the f32 form is no longer exact."""
    x, y = cases(128 * 18)
    big = np.abs(np.abs(x) - np.abs(y)) > 2**23
    assert np.count_nonzero(big) > 1000
    try:
        got = disc_f32_model(x, y, 0)
    except AssertionError:
        return                                                  # x / y / den no longer exact in f32: equally disqualifying
    assert np.count_nonzero(got != fast_atan2_ref(y, x)) > 0


def test_kernel_limit_matches_this_model():
    import os, re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rtl-sdr-rs_amd", "csrc", "fmd_device.h")).read()
    assert int(re.search(r"#define FMD_DISC_F32_MAX_D (\d+)", hdr).group(1)) * 128 == LIM


def test_low_16_bits_of_the_biased_result_are_the_i16():
    """disc_f32_xy<.., LO16>: the kernels store `res + 1.5 * 2^23` truncated to 16 bits instead of converting res to i32 first."""
    res = np.arange(-16384, 16385, dtype=np.int64).astype(F)
    bits = (res + F(12582912.0)).view(np.uint32)
    assert np.array_equal((bits & np.uint32(0xFFFF)).astype(np.uint16).view(np.int16), res.astype(np.int16))
    nan = np.array([0x7FC00000, 0xFFC00000], dtype=np.uint32).view(F)          # the (0, 0) case: the hardware's quiet NaN, either sign
    assert np.array_equal((nan + F(12582912.0)).view(np.uint32) & np.uint32(0xFFFF), np.zeros(2, np.uint32))
