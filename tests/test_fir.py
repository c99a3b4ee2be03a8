"""Row G' (SURVEY 8a): generalised tapped decimating FIR.  No reference counterpart exists, so the anchor is the
reduction the survey demands: with taps = 1...1 and n_taps == decim == downsample the FIR must reproduce
Demod::low_pass_complex (simple_fm.rs:337-352) exactly, including its phase/partial-sum carry across calls.
CPU: the oracle FIR against the oracle boxcar and against a numpy convolution.  GPU: HIP FIR against the oracle FIR."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from conftest import run_in_exp_child


def rotated_stream(oracle, data):
    """rotate_90 + centre + pair of a whole stream (call lengths are multiples of 8 bytes, so per-call rotation
    equals whole-stream rotation)."""
    buf = np.ascontiguousarray(data, dtype=np.uint8).copy()
    assert oracle.lib.fmo_rotate_90(buf.ctypes.data_as(C.POINTER(C.c_uint8)), buf.size) == 0
    s = buf.astype(np.int64) - 127
    return s[0::2], s[1::2]


def boxcar_oracle(oracle, D, chunks):
    d = oracle.new(oracle.config(D, 48000, 48000))
    outs = []
    for ch in chunks:
        buf = np.ascontiguousarray(ch, dtype=np.uint8).copy()
        oracle.lib.fmo_rotate_90(buf.ctypes.data_as(C.POINTER(C.c_uint8)), buf.size)
        sig = np.empty(buf.size, dtype=np.int16)
        oracle.lib.fmo_center(buf.ctypes.data_as(C.POINTER(C.c_uint8)), buf.size, sig.ctypes.data_as(C.POINTER(C.c_int16)))
        cplx = (oracle_lib.Cplx * (buf.size // 2))()
        n = oracle.lib.fmo_buf_to_complex(sig.ctypes.data_as(C.POINTER(C.c_int16)), sig.size, cplx)
        out = (oracle_lib.Cplx * (n // D + 2))()
        m = oracle.lib.fmo_low_pass_complex(C.byref(d), cplx, n, out)
        outs.extend([out[i].re, out[i].im] for i in range(m))
    return np.array(outs, dtype=np.int64).reshape(-1, 2)


@pytest.mark.parametrize("D", [2, 6, 10, 16])
def test_oracle_fir_all_ones_is_low_pass_complex(oracle, D):
    rng = np.random.default_rng(D)
    chunks = [rng.integers(0, 256, int(n) * 8, dtype=np.uint8) for n in rng.integers(1, 60, 6)]
    h = oracle.fir_new(np.ones(D, np.int16), D)
    got = np.concatenate([oracle.fir_filter(h, ch) for ch in chunks])
    oracle.lib.fmo_fir_free(h)
    exp = boxcar_oracle(oracle, D, chunks)
    assert got.shape == exp.shape and np.array_equal(got, exp)


@pytest.mark.parametrize("T,M", [(127, 8), (5, 2), (16, 16), (33, 4), (1, 2), (64, 6)])
def test_oracle_fir_equals_numpy_convolution(oracle, T, M):
    rng = np.random.default_rng(T * 100 + M)
    taps = rng.integers(-2047, 2048, T).astype(np.int16)
    chunks = [rng.integers(0, 256, int(n) * 8, dtype=np.uint8) for n in rng.integers(1, 120, 5)]
    h = oracle.fir_new(taps, M)
    got = np.concatenate([oracle.fir_filter(h, ch) for ch in chunks] + [np.empty((0, 2), np.int32)])
    oracle.lib.fmo_fir_free(h)
    re, im = rotated_stream(oracle, np.concatenate(chunks))
    n_out = (re.size - T) // M + 1 if re.size >= T else 0
    exp = np.array([[int(np.dot(taps.astype(np.int64), re[M * m: M * m + T])),
                     int(np.dot(taps.astype(np.int64), im[M * m: M * m + T]))] for m in range(n_out)], dtype=np.int64).reshape(-1, 2)
    assert got.shape == exp.shape and np.array_equal(got, exp)


FIR_SHAPES = [(127, 8), (6, 6), (10, 10), (5, 2), (16, 16), (33, 4), (1, 2), (64, 6), (255, 32),
              (300, 8), (1024, 2), (129, 64), (77, 66), (31, 128),
              (63, 8), (200, 8), (64, 16), (31, 12), (100, 10), (9, 26)]      # 12-bit taps at decim >= 8 in one K pass: the sparse two-digit form


def fir_stream_case(fmd, oracle, T, M):
    """One FIR shape streamed over ragged calls against the oracle FIR."""
    rng = np.random.default_rng(T * 7 + M)
    taps = np.ones(T, np.int16) if T == M else rng.integers(-2047, 2048, T).astype(np.int16)
    nch = 5
    bank = fmd.FirBank(taps, M, nch)
    hs = [oracle.fir_new(taps, M) for _ in range(nch)]
    for call in range(5):
        n = int(rng.integers(1, 400)) * 8 if call else 8          # a first call too short to emit anything
        if call == 3:
            n = 8 * int(rng.integers(6000, 9000))                 # several tiles per channel
        iq = rng.integers(0, 256, (nch, n), dtype=np.uint8)
        if call == 2:
            iq[:] = np.where(rng.integers(0, 2, (nch, n)) > 0, 255, 0)   # full scale
        got = bank.filter_batch(iq)
        for c in range(nch):
            exp = oracle.fir_filter(hs[c], iq[c])
            assert got[c].shape == exp.shape, (T, M, call, c, got[c].shape, exp.shape)
            assert np.array_equal(got[c], exp), (T, M, call, c)
    for h in hs:
        oracle.lib.fmo_fir_free(h)
    bank.reset()
    assert bank.filter_batch(np.zeros((nch, 8), np.uint8)).shape[1] == ((4 - T) // M + 1 if T <= 4 else 0)   # after reset


@pytest.mark.gpu
@pytest.mark.parametrize("T,M", FIR_SHAPES)
def test_gpu_fir_matches_oracle(fmd, oracle, T, M):
    """What the shipped library selects by itself: the matrix-core form for decim <= 64, the VALU form beyond; shapes cover
    one and several K passes, both window parities, decim > 64."""
    fir_stream_case(fmd, oracle, T, M)


def fir_stream_case_8bit(fmd, oracle, T, M, want_digits):
    """The same streaming walk with an 8-bit filter (every |tap| <= 127; the extremes +-127 among them)."""
    rng = np.random.default_rng(T * 11 + M)
    taps = rng.integers(-127, 128, T).astype(np.int16)
    taps[rng.integers(0, T)] = 127
    taps[rng.integers(0, T)] = -127
    nch = 3
    bank = fmd.FirBank(taps, M, nch)
    assert bank.tap_digits() == want_digits, (T, M, bank.tap_digits())
    hs = [oracle.fir_new(taps, M) for _ in range(nch)]
    for call in range(5):
        n = int(rng.integers(1, 400)) * 8 if call else 8
        if call == 3:
            n = 8 * int(rng.integers(6000, 9000))                 # several tiles per channel
        iq = rng.integers(0, 256, (nch, n), dtype=np.uint8)
        if call == 2:
            iq[:] = np.where(rng.integers(0, 2, (nch, n)) > 0, 255, 0)   # full scale
        got = bank.filter_batch(iq)
        for c in range(nch):
            exp = oracle.fir_filter(hs[c], iq[c])
            assert got[c].shape == exp.shape and np.array_equal(got[c], exp), (T, M, call, c)
    for h in hs:
        oracle.lib.fmo_fir_free(h)


@pytest.mark.gpu
@pytest.mark.parametrize("T,M", FIR_SHAPES)
def test_gpu_fir_one_digit_form_matches_oracle(fmd, oracle, T, M):
    """Every |tap| <= 127: the matrix-core kernel takes ONE i8 digit per tap and eight outputs per operand column
    (fmd_fir_mfma_kernel<NKU, SWZ, 1>); decim > 64 keeps the vector-pipe kernel.  Same shapes as the two-digit test: one and several K
    passes, both window parities (an odd decim / 2 gives the two outputs of a lane different additive constants)."""
    fir_stream_case_8bit(fmd, oracle, T, M, 1 if M <= 64 else 0)


@pytest.mark.gpu
def test_gpu_fir_kernel_name_follows_the_form(fmd):
    """fmd_fir_kernel_name (round 6): the name rocprofv3 --kernel-trace prints for the handle's launches -- what bench.py quotes in
    `extra.config4_fir.kernel` instead of a constant -- follows the form the taps select."""
    rng = np.random.default_rng(2)
    wide = rng.integers(-2047, 2048, 127).astype(np.int16)
    assert fmd.FirBank(wide, 8).kernel_name() == "(anonymous namespace)::fmd_fir_mfma_kernel<6, false, 3>"      # config 4: sparse two-digit form
    assert fmd.FirBank(np.clip(wide, -127, 127), 8).kernel_name() == "(anonymous namespace)::fmd_fir_mfma_kernel<6, false, 1>"
    assert fmd.FirBank(wide, 4).kernel_name().endswith(", false, 2>")                                           # decim < 8: dense two digits
    assert fmd.FirBank(np.ones(4, np.int16), 128).kernel_name() == "(anonymous namespace)::fmd_fir_kernel"       # vector-pipe kernel


@pytest.mark.gpu
def test_gpu_fir_digits_follow_the_taps(fmd):
    assert fmd.FirBank(np.full(9, 127, np.int16), 8).tap_digits() == 1
    assert fmd.FirBank(np.array([5, -128, 3], np.int16), 8).tap_digits() == 2      # -128 is one too many for the sign-flipped rows
    assert fmd.FirBank(np.array([5, 2047, 3], np.int16), 8).tap_digits() == 2
    assert fmd.FirBank(np.ones(4, np.int16), 128).tap_digits() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["valu", "mfma_swz", "two_digits", "dense"])
def test_gpu_fir_forced_forms(fmd, oracle, request, form):
    """The VALU form forced for small decim (FMD_FIR_MFMA=0), the conflict-free LDS layout of the dense matrix-core form
    (FMD_FIR_SWZ=1, decim 8 only), an 8-bit filter through two digits (FMD_FIR_DIGITS=2) and the dense interleaved two-digit form where
    the library now takes the sparse one (FMD_FIR_SPARSE=0) are knobs of the -DFMD_EXPERIMENT build: each form re-runs itself ONCE in a child process
    on that library and walks through all its shapes there."""
    if run_in_exp_child(request, {"FMD_FIR_MFMA": "0"} if form == "valu" else {"FMD_FIR_SWZ": "1", "FMD_FIR_SPARSE": "0"} if form == "mfma_swz"
                        else {"FMD_FIR_DIGITS": "2"} if form == "two_digits" else {"FMD_FIR_SPARSE": "0"}):
        return
    for T, M in FIR_SHAPES:
        if form == "mfma_swz" and M != 8:
            continue
        if form == "two_digits":                                  # an 8-bit filter forced through the two-digit form (the A/B knob)
            fir_stream_case_8bit(fmd, oracle, T, M, 2 if M <= 64 else 0)
            continue
        fir_stream_case(fmd, oracle, T, M)
        if form == "mfma_swz":
            fir_stream_case_8bit(fmd, oracle, T, M, 1)            # the swizzled layout under the one-digit form (decim 8: columns 128 bytes apart: plain)


@pytest.mark.gpu
def test_gpu_fir_config4_full_size(fmd, oracle):
    """BASELINE configs[3] at its stated size: 127 taps, decimate by 8, 256 channels x 2 MiB per call (52.4 ms at
    20 Msps), two consecutive calls (history carry), EVERY channel compared -- this is the launch the config-4 bench
    number comes from, incl. its XCD-aware grid (8, tiles, 32) and the c = blockIdx.x * gridDim.z + blockIdx.z map."""
    rng = np.random.default_rng(4)
    taps = rng.integers(-2047, 2048, 127).astype(np.int16)
    nch, n = 256, 2 << 20
    bank = fmd.FirBank(taps, 8, nch)
    hs = [oracle.fir_new(taps, 8) for _ in range(nch)]
    for call in range(2):
        iq = fmd.synth.synth_iq(nch, n, sample_offset=call * (n // 2), amplitude=110)
        if call == 1:
            iq[5::17, ::2] = 255; iq[5::17, 1::2] = 0              # some full-scale channels
        got = bank.filter_batch(iq)
        exp = oracle.fir_filter_batch(hs, iq, cap=n // 16 + 8)
        assert got.shape == exp.shape and got.shape[1] in (131057, 131072)
        bad = [c for c in range(nch) if not np.array_equal(got[c], exp[c])]
        assert not bad, (call, bad[:8])
    for h in hs:
        oracle.lib.fmo_fir_free(h)


def test_oracle_fir_batch_equals_single(oracle):
    rng = np.random.default_rng(12)
    taps = rng.integers(-2047, 2048, 33).astype(np.int16)
    hs, hs1 = [oracle.fir_new(taps, 4) for _ in range(5)], [oracle.fir_new(taps, 4) for _ in range(5)]
    for _ in range(3):
        iq = rng.integers(0, 256, (5, 8 * int(rng.integers(20, 200))), dtype=np.uint8)
        got = oracle.fir_filter_batch(hs, iq, threads=3)
        for c in range(5):
            assert np.array_equal(got[c], oracle.fir_filter(hs1[c], iq[c]))
    for h in hs + hs1:
        oracle.lib.fmo_fir_free(h)


@pytest.mark.gpu
def test_gpu_fir_errors(fmd):
    with pytest.raises(fmd.FmdError):
        fmd.FirBank(np.ones(4, np.int16), 3)                       # odd decimation
    with pytest.raises(fmd.FmdError):
        fmd.FirBank(np.full(4, 4000, np.int16), 2)                 # |tap| > 2047
    b = fmd.FirBank(np.ones(4, np.int16), 2)
    with pytest.raises(fmd.FmdError) as ei:
        b.filter_batch(np.zeros((1, 12), np.uint8))
    assert ei.value.status == -2


@pytest.mark.gpu
def test_gpu_fir_fuzz(fmd, oracle):
    """Random taps / decimation / call sizes, streaming, both kernel forms chosen by the library's own rule
    (matrix-core form for decim <= 64, VALU form beyond).  FMD_FUZZ_CASES scales the number of cases."""
    import os
    n_cases = int(os.environ.get("FMD_FUZZ_CASES", "40"))
    rng = np.random.default_rng(int(os.environ.get("FMD_FUZZ_SEED", "4242")))
    for _ in range(n_cases):
        M = 2 * int(rng.choice([1, 2, 3, 4, 5, 8, 16, 25, 32, 33, 40]))
        T = int(rng.choice([1, 2, 3, 7, 16, 31, 64, 127, 128, 200, 513, 1024]))
        lim = 127 if rng.integers(0, 3) == 0 else 2047                 # a third of the cases: an 8-bit filter (the one-digit form)
        taps = rng.integers(-lim, lim + 1, T).astype(np.int16)
        nch = int(rng.integers(1, 5))
        bank = fmd.FirBank(taps, M, nch)
        assert bank.tap_digits() == (0 if M > 64 else 1 if np.abs(taps).max() <= 127 else 2), (T, M, bank.tap_digits())
        hs = [oracle.fir_new(taps, M) for _ in range(nch)]
        for _ in range(int(rng.integers(1, 5))):
            n = 8 * int(rng.integers(1, 3000))
            iq = rng.integers(0, 256, (nch, n), dtype=np.uint8)
            got = bank.filter_batch(iq)
            for c in range(nch):
                exp = oracle.fir_filter(hs[c], iq[c])
                assert got[c].shape == exp.shape and np.array_equal(got[c], exp), (T, M, n, c)
        for h in hs:
            oracle.lib.fmo_fir_free(h)
        bank.close() if hasattr(bank, "close") else None
