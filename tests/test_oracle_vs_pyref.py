"""Cross-check the C oracle against the independent Python restatement on the behaviours the
reference's own tests do not cover: rotate_90, centring, end-to-end
demodulate, state carry across calls, the i32 wrap in fast_atan2, the f64 sample with a
non-zero predecessor, error behaviour."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
import pyref


def test_rotate_90_pattern(oracle):
    buf = np.arange(64, dtype=np.uint8) * 3 + 1
    b = buf.copy()
    assert oracle.lib.fmo_rotate_90(b.ctypes.data_as(C.POINTER(C.c_uint8)), b.size) == 0
    exp = buf.copy().reshape(-1, 8)
    exp = np.stack([exp[:, 0], exp[:, 1], 255 - exp[:, 3], exp[:, 2], 255 - exp[:, 4], 255 - exp[:, 5],
                    exp[:, 7], 255 - exp[:, 6]], axis=1).astype(np.uint8).reshape(-1)
    assert np.array_equal(b, exp)                      # doc comment simple_fm.rs:275
    assert bytes(b) == pyref.rotate_90(bytes(buf))
    bad = np.zeros(12, dtype=np.uint8)
    assert oracle.lib.fmo_rotate_90(bad.ctypes.data_as(C.POINTER(C.c_uint8)), 12) == -1


def test_centring_is_minus_127(oracle):
    buf = np.array([0, 1, 127, 128, 254, 255, 7, 9], dtype=np.uint8)
    out = np.empty(8, dtype=np.int16)
    oracle.lib.fmo_center(buf.ctypes.data_as(C.POINTER(C.c_uint8)), 8, out.ctypes.data_as(C.POINTER(C.c_int16)))
    assert out.tolist() == [-127, -126, 0, 1, 127, 128, -120, -118]


@pytest.mark.parametrize("y,x,exp", [(0, 1179648, 3641), (0, 524287, 0), (0, 524288, 8192), (0, 0, 0)])
def test_fast_atan2_wrap_points(oracle, y, x, exp):
    """the i64 product is truncated to i32 before the divide."""
    assert oracle.lib.fmo_fast_atan2(y, x) == exp
    assert pyref.fast_atan2(y, x) == exp
    assert oracle.lib.fmcf_fast_atan2(y, x) == exp


def test_fast_atan2_random_agreement(oracle):
    rng = np.random.default_rng(7)
    ys = np.concatenate([rng.integers(-1_200_000, 1_200_001, 4000), rng.integers(-300, 301, 4000)])
    xs = np.concatenate([rng.integers(-1_200_000, 1_200_001, 4000), rng.integers(-300, 301, 4000)])
    for y, x in zip(ys.tolist(), xs.tolist()):
        r = oracle.lib.fmo_fast_atan2(y, x)
        assert r == pyref.fast_atan2(y, x) == oracle.lib.fmcf_fast_atan2(y, x)
        assert -16384 <= r <= 16384                   # always fits the `as i16`


def test_polar_discriminant_f64_nonzero_predecessor(oracle):
    rng = np.random.default_rng(11)
    for _ in range(2000):
        a = tuple(int(v) for v in rng.integers(-1280, 1281, 2))
        b = tuple(int(v) for v in rng.integers(-1280, 1281, 2))
        assert oracle.lib.fmo_polar_discriminant(oracle_lib.Cplx(*a), oracle_lib.Cplx(*b)) == \
            pyref.polar_discriminant(a, b)


@pytest.mark.parametrize("D,fast,slow", [(6, 170000, 32000), (10, 240000, 32000), (7, 166666, 32000),
                                         (1, 48000, 48000), (5, 250000, 44100)])
def test_demodulate_end_to_end_and_state_carry(oracle, D, fast, slow):
    rng = np.random.default_rng(D * 1000 + 3)
    cfg = oracle.config(D, fast, slow)
    d = oracle.new(cfg)
    pd = pyref.Demod(D, fast, slow)
    for call in range(4):
        n = int(rng.integers(4, 60)) * 8 + 8 * D
        buf = rng.integers(0, 256, n, dtype=np.uint8)
        if call == 2:
            buf[: n // 2] = np.tile(np.array([255, 255, 0, 255, 0, 0, 255, 0], dtype=np.uint8), n // 16 + 1)[: n // 2]
        got = oracle.demodulate(d, buf)
        exp = pd.demodulate(buf.tobytes())
        assert got.tolist() == exp
        assert oracle.state_of(d) == pd.state()


def test_demodulate_errors(oracle):
    d = oracle.new(oracle.config(6, 170000, 32000))
    out = np.empty(64, dtype=np.int16)
    op = out.ctypes.data_as(C.POINTER(C.c_int16))
    b = np.zeros(64, dtype=np.uint8)
    bp = b.ctypes.data_as(C.POINTER(C.c_uint8))
    assert oracle.lib.fmo_demodulate(C.byref(d), bp, 12, op, 64) == -1   # len % 8
    assert oracle.lib.fmo_demodulate(C.byref(d), bp, 16, op, 64) == -2   # 8 samples / 6 -> 1 decimated
    d = oracle.new(oracle.config(6, 170000, 32000))
    assert oracle.lib.fmo_demodulate(C.byref(d), bp, 24, op, 64) == 0    # 2 decimated, no audio yet
    with pytest.raises(AssertionError):
        pyref.Demod(6, 170000, 32000).demodulate(bytes(16))


def test_file_mode_complete_blocks_only(oracle):
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, 3 * 4096 + 1000, dtype=np.uint8)
    cfg = oracle.config(6, 170000, 32000)
    d = oracle.new(cfg)
    out = np.empty(data.size, dtype=np.int16)
    n = oracle.lib.fmo_file_mode(C.byref(d), data.ctypes.data_as(C.POINTER(C.c_uint8)), data.size, 4096,
                                 out.ctypes.data_as(C.POINTER(C.c_int16)), out.size)
    exp, _ = oracle.demodulate_stream(cfg, data[: 3 * 4096], 4096)
    assert n == exp.size and np.array_equal(out[:n], exp)
