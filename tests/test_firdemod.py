"""Tapped FIR -> discriminator -> resampler fused in one kernel (fmd_firdemod_*, BASELINE north_star's "FIR + demod +
resample fused").  No reference counterpart exists for a tapped FIR, so the anchors are:
  * CPU: the oracle's composition fmo_firdemod_* (fmo_fir_filter -> fmo_fm_demod -> fmo_low_pass_real) with all-ones taps,
    n_taps == decim == downsample, shift 0 IS fmo_demodulate -- the reference chain (simple_fm.rs:256-269), audio and state;
  * GPU: the fused kernel == that composition for arbitrary taps / shifts / rates / ragged streaming calls, and therefore
    == the boxcar kernel == the reference for all-ones taps (checked directly too), incl. BASELINE configs[3] at its full
    size (127 taps, decimate 8, 256 channels x 2 MiB) with every channel compared."""
import math

import numpy as np
import pytest


@pytest.mark.parametrize("D,fast,slow", [(6, 170000, 32000), (10, 240000, 32000), (2, 48000, 48000), (16, 150000, 32000)])
def test_oracle_composition_reduces_to_reference_chain(oracle, D, fast, slow):
    rng = np.random.default_rng(D)
    h = oracle.firdemod_new(np.ones(D, np.int16), D, 0, fast, slow)
    d = oracle.new(oracle.config(D, fast, slow))
    for i in range(6):
        n = 8 * int(rng.integers(4 * D, 600))
        b = rng.integers(0, 256, n, dtype=np.uint8) if i % 2 else np.where(rng.integers(0, 2, n) > 0, 255, 0).astype(np.uint8)
        assert np.array_equal(oracle.firdemod(h, b), oracle.demodulate(d, b))
    s1, s2 = oracle.firdemod_state(h), oracle.state_of(d)
    assert (s1["now_lpr"], s1["prev_lpr_index"], s1["demod_pre"]) == (s2["now_lpr"], s2["prev_lpr_index"], s2["demod_pre"])
    oracle.lib.fmo_firdemod_free(h)


def state_tuple(s):
    return (s["now_lpr"], s["prev_lpr_index"], s["demod_pre"])


@pytest.mark.gpu
@pytest.mark.parametrize("D,fast,slow", [(6, 170000, 32000), (10, 240000, 32000), (2, 48000, 48000), (16, 150000, 32000),
                                         (64, 37500, 8000), (8, 250000, 8000)])      # the last one: the register form (decimate 8)
def test_gpu_all_ones_is_the_boxcar_kernel_and_the_reference(fmd, oracle, D, fast, slow):
    """taps = 1...1, n_taps = decim = downsample, shift 0: fused FIR kernel == boxcar kernel == oracle of the reference."""
    rng = np.random.default_rng(D + 100)
    nch = 6
    fd = fmd.FirDemodBank(np.ones(D, np.int16), D, fast, slow, nch, shift=0)
    cfg = fmd.DemodConfig(fast, fast, slow, D, max(1, (1 << 15) // (128 * D)))
    bank = fmd.DemodBank(cfg, nch)
    ods = [oracle.new(oracle.config(D, fast, slow)) for _ in range(nch)]
    for call in range(5):
        n = 8 * int(rng.integers(4 * D, 4000)) if call != 2 else fmd.DEFAULT_BUF_LENGTH
        iq = rng.integers(0, 256, (nch, n), dtype=np.uint8)
        if call == 3:
            iq[:] = np.where(rng.integers(0, 2, (nch, n)) > 0, 255, 0)           # full scale: the fast_atan2 wrap
        got = fd.demodulate_batch(iq)
        box = bank.demodulate_batch(iq)
        for c in range(nch):
            exp = oracle.demodulate(ods[c], iq[c])
            assert np.array_equal(got[c], exp), (call, c)
            assert np.array_equal(box[c], exp), (call, c)
    for c in (0, nch - 1):
        assert state_tuple(fd.get_state(c).as_dict()) == state_tuple(oracle.state_of(ods[c]))


@pytest.mark.gpu
@pytest.mark.parametrize("T,M,fast,slow", [(127, 8, 2500000, 48000), (33, 4, 250000, 48000), (5, 2, 96000, 48000),
                                           (64, 6, 170000, 32000), (255, 32, 625000, 8000), (300, 8, 100000, 44100),
                                           (16, 16, 48000, 48000), (129, 64, 37500, 8000), (1, 2, 500000, 32000),
                                           (127, 16, 1250000, 48000), (200, 16, 625000, 44100), (255, 16, 625000, 48000)])
def test_gpu_fused_matches_composition(fmd, oracle, T, M, fast, slow):
    """Decimate 8 with all k-steps in one pass runs the operand-fragment-reuse mapping; every other shape the plain one."""
    fused_case(fmd, oracle, T, M, fast, slow)


@pytest.mark.gpu
def test_gpu_fused_plain_mapping_forced(fmd, oracle, request):
    """The plain matrix-core mapping where the shipped library picks fragment reuse (FMD_FD_NOREUSE, a knob of the
    -DFMD_EXPERIMENT build: the test re-runs itself once in a child process on that library)."""
    from conftest import run_in_exp_child
    if run_in_exp_child(request, {"FMD_FD_NOREUSE": "1"}):
        return
    for T, M, fast, slow in [(127, 8, 2500000, 48000), (127, 16, 1250000, 48000), (200, 16, 625000, 44100)]:
        fused_case(fmd, oracle, T, M, fast, slow)


@pytest.mark.gpu
def test_gpu_fused_reuse_at_decim_16(fmd, oracle, request):
    """Fragment reuse with two k-steps between a column's output groups (decim 16; FMD_FD_REUSE16, experiment build --
    bit-exact, measured 4-6 % slower than the plain mapping, so the shipped library does not select it) and the
    even-column-pitch form of the decim-8 mapping (FMD_DBG bit 8)."""
    from conftest import run_in_exp_child
    if run_in_exp_child(request, {"FMD_FD_REUSE16": "1", "FMD_DBG": "256"}):
        return
    for T, M, fast, slow in [(127, 16, 1250000, 48000), (200, 16, 625000, 44100), (33, 16, 312500, 48000), (127, 8, 2500000, 48000)]:
        fused_case(fmd, oracle, T, M, fast, slow)


def reg_kernel_prefix(fast, slow, taps_8bit, knobs=None):
    """Which kernel the library reports for a decimate-8 bank with the f32 discriminator (csrc/fmd_firdemod.hip): the register form
    needs audio groups of >= 4 NG outputs (NG = the longest column parameter <= 8 they admit, or FMD_FD_REG); an even NG runs the
    matrix phase on the 4:2 sparse instruction (`s`; FMD_FD_SPARSE=0: dense), an 8-bit filter with an even NG one tap digit (`1`)."""
    knobs = knobs or {}
    fa = fast // slow
    ng = int(knobs["FMD_FD_REG"]) if "FMD_FD_REG" in knobs else min(fa // 4, 8)
    if taps_8bit and ng in (5, 7) and "FMD_FD_REG" not in knobs and knobs.get("FMD_FD_SPARSE") != "0" and knobs.get("FMD_FD_DIGITS") != "2":
        ng -= 1                                                   # an 8-bit filter takes the even column parameter below an odd one
    pre = "(anonymous namespace)::fmd_firdemod_"
    if ng < 4 or fa < 4 * ng:
        return pre + "kernel<"
    even = ng in (4, 6, 8)
    one = taps_8bit and even and knobs.get("FMD_FD_DIGITS") != "2"
    sparse = even and knobs.get("FMD_FD_SPARSE") != "0"
    return pre + "reg" + ("1" if one else "") + ("s" if sparse else "") + "_kernel<"


REG_SHAPES = [(127, 8, 2500000, 48000), (127, 8, 1000000, 44100), (8, 8, 250000, 8000), (200, 8, 480000, 8000), (33, 8, 960000, 48000),
              (1, 8, 640000, 32000)]


@pytest.mark.gpu
@pytest.mark.parametrize("T,M,fast,slow", REG_SHAPES)
def test_gpu_fused_register_form(fmd, oracle, T, M, fast, slow):
    """Decimate 8 with the f32 discriminator (|lp| <= 2048) and audio groups of >= 20 filter outputs: the form that takes
    the discriminator straight out of the matrix-core result registers (fmd_firdemod_reg_kernel): tiny first calls, many
    tiles per channel, full scale, state carried over six calls."""
    kn = fused_case(fmd, oracle, T, M, fast, slow, f32_only=True)
    assert kn.startswith(reg_kernel_prefix(fast, slow, False)) and "reg" in kn and kn.endswith(", true>"), kn


@pytest.mark.gpu
@pytest.mark.parametrize("ng", ["0", "4", "6", "10"])
def test_gpu_fused_register_form_variants(fmd, oracle, request, ng):
    """The same shapes with 4 / 6 / 10 output groups per column (columns of 14 / 22 / 38 outputs; the default is the longest the
    audio groups admit up to 8) and with the form switched off (the
    LDS-array kernel that every other decimation runs): knobs of the -DFMD_EXPERIMENT build."""
    from conftest import run_in_exp_child
    if run_in_exp_child(request, {"FMD_FD_REG": ng}):
        return
    for T, M, fast, slow in REG_SHAPES[:4]:
        kn = fused_case(fmd, oracle, T, M, fast, slow, f32_only=True)
        want_reg = ng != "0" and fast // slow >= 4 * int(ng)         # an audio group must hold a lane's 4 ng - 3 consecutive outputs
        assert kn.startswith(reg_kernel_prefix(fast, slow, False, {"FMD_FD_REG": ng})) and ("reg" in kn) == want_reg, kn
        assert not want_reg or kn.endswith(", %s, true>" % ng), kn


REG1_SHAPES = [(127, 8, 2500000, 48000), (200, 8, 480000, 8000), (64, 8, 768000, 48000), (100, 8, 1200000, 48000), (8, 8, 256000, 8000),
               (1, 8, 1280000, 32000), (127, 8, 1000000, 44100), (33, 8, 960000, 32000)]   # column parameter NG = 8, 8, 4, 6, 8, 8, 5 -> 4, 7 -> 6


@pytest.mark.gpu
@pytest.mark.parametrize("T,M,fast,slow", REG1_SHAPES)
def test_gpu_fused_register_form_one_digit(fmd, oracle, T, M, fast, slow):
    """An 8-bit filter (every |tap| <= 127, +-127 among them) in the register form with an even column parameter: one i8 digit per tap,
    (re, im) of eight outputs per operand fragment, half the accumulators, the matrix phase on the 4:2 sparse instruction
    (fmd_firdemod_reg1s_kernel).  Same walk as the two-digit test:
    tiny first calls, many tiles per channel, full scale, state carried over six calls."""
    kn = fused_case(fmd, oracle, T, M, fast, slow, f32_only=True, taps_max=127)
    assert kn.startswith("(anonymous namespace)::fmd_firdemod_reg1s_kernel<") and kn.endswith(", true>"), kn
    assert kn.startswith(reg_kernel_prefix(fast, slow, True)), kn


@pytest.mark.gpu
@pytest.mark.parametrize("knobs", [{"FMD_FD_DIGITS": "2"}, {"FMD_FD_REG": "4"}, {"FMD_FD_REG": "6"}, {"FMD_FD_REG": "5"}, {"FMD_FD_ROWS": "0"},
                                   {"FMD_FD_SPARSE": "0"}, {"FMD_FD_SPARSE": "0", "FMD_FD_REG": "6"}, {"FMD_FD_SPARSE": "0", "FMD_FD_DIGITS": "2"}])
def test_gpu_fused_digit_and_sparse_variants(fmd, oracle, request, knobs):
    """8-bit and 12-bit filters side by side with the two-digit form forced, with shorter columns (4 / 6: sparse, one digit for the 8-bit
    filter; 5: odd, so dense and two digits), without the per-tile table, and with the matrix phase on the dense instruction
    (FMD_FD_SPARSE=0): knobs of the -DFMD_EXPERIMENT build."""
    from conftest import run_in_exp_child
    if run_in_exp_child(request, knobs):
        return
    for T, M, fast, slow in REG1_SHAPES[:3]:
        kn = fused_case(fmd, oracle, T, M, fast, slow, f32_only=True, taps_max=127)
        kn2 = fused_case(fmd, oracle, T, M, fast, slow, f32_only=True)                       # 12-bit taps: two digits
        assert kn.startswith(reg_kernel_prefix(fast, slow, True, knobs)), (knobs, kn)
        assert kn2.startswith(reg_kernel_prefix(fast, slow, False, knobs)), (knobs, kn2)


def fused_case(fmd, oracle, T, M, fast, slow, f32_only=False, taps_max=2047):
    rng = np.random.default_rng(T * 11 + M)
    taps = rng.integers(-taps_max, taps_max + 1, T).astype(np.int16)
    if taps_max < 2047:
        taps[rng.integers(0, T)] = taps_max
        taps[rng.integers(0, T)] = -taps_max
    shift = fmd.auto_shift(taps, 16384) + int(rng.integers(0, 6))     # both discriminator forms (|lp| <= 2048: f32)
    if f32_only:
        shift = fmd.auto_shift(taps, 2048) + int(rng.integers(0, 3))
    nch = 5
    fd = fmd.FirDemodBank(taps, M, fast, slow, nch, shift=shift)
    hs = [oracle.firdemod_new(taps, M, shift, fast, slow) for _ in range(nch)]
    first = 8 * ((T + 2 * M) // 4 + 2)                                            # >= 2 filter outputs
    for call in range(6):
        n = first if call == 0 else 8 * int(rng.integers(2 * M, 700))
        if call == 3:
            n = 8 * int(rng.integers(6000, 9000))                                  # several tiles per channel
        iq = rng.integers(0, 256, (nch, n), dtype=np.uint8)
        if call == 4:
            iq[:] = np.where(rng.integers(0, 2, (nch, n)) > 0, 255, 0)
        got = fd.demodulate_batch(iq)
        for c in range(nch):
            exp = oracle.firdemod(hs[c], iq[c])
            assert got[c].shape == exp.shape, (call, c, got[c].shape, exp.shape)
            assert np.array_equal(got[c], exp), (call, c)
    for c in range(nch):
        assert state_tuple(fd.get_state(c).as_dict()) == state_tuple(oracle.firdemod_state(hs[c]))
        oracle.lib.fmo_firdemod_free(hs[c])
    fd.reset()
    assert fd.get_state(0).as_dict()["demod_pre"] == [0, 0]
    return fd.kernel_name()


@pytest.mark.gpu
def test_gpu_fused_config4_full_size(fmd, oracle):
    """BASELINE configs[3] shape with the demodulator behind it: 127 taps, decimate 8, 256 channels x 2 MiB per call,
    2.5 Msps -> 48 kHz; two consecutive calls, EVERY channel compared (the launch bench.py's extra line times)."""
    import torch
    rng = np.random.default_rng(1)
    taps = rng.integers(-2047, 2048, 127).astype(np.int16)
    nch, n, fast, slow = 256, 2 << 20, 2500000, 48000
    fd = fmd.FirDemodBank(taps, 8, fast, slow, nch)
    hs = [oracle.firdemod_new(taps, 8, fd.shift, fast, slow) for _ in range(nch)]
    cap = fd.out_cap(n)
    d_out = torch.zeros((nch, cap), dtype=torch.int16, device="cuda")
    for call in range(2):
        iq = fmd.synth.synth_iq(nch, n, sample_offset=call * (n // 2), amplitude=110)
        d_iq = torch.from_numpy(iq).cuda()
        k = fd.demodulate_device(d_iq.data_ptr(), n, d_out.data_ptr(), cap)
        fd.check()
        exp, lens = oracle.firdemod_batch(hs, iq, cap)
        assert int(lens.min()) == int(lens.max()) == k and k > 2500
        got = d_out.cpu().numpy()
        bad = [c for c in range(nch) if not np.array_equal(got[c, :k], exp[c, :k])]
        assert not bad, (call, bad[:8])
    for h in hs:
        oracle.lib.fmo_firdemod_free(h)


@pytest.mark.gpu
def test_gpu_fused_errors(fmd):
    with pytest.raises(fmd.FmdError) as ei:
        fmd.FirDemodBank(np.full(127, 2047, np.int16), 8, 2500000, 48000, shift=0)     # gain beyond the discriminator's range
    assert ei.value.status == -6
    with pytest.raises(fmd.FmdError):
        fmd.FirDemodBank(np.ones(4, np.int16), 3, 48000, 48000)                        # odd decimation
    with pytest.raises(fmd.FmdError) as ei:
        fmd.FirDemodBank(np.ones(4, np.int16), 4, 32000, 48000)                        # rate_out < rate_resample
    assert ei.value.status == -4
    b = fmd.FirDemodBank(np.ones(4, np.int16), 4, 48000, 48000)
    with pytest.raises(fmd.FmdError) as ei:
        b.demodulate_batch(np.zeros((1, 12), np.uint8))
    assert ei.value.status == -2
    with pytest.raises(fmd.FmdError) as ei:
        b.demodulate_batch(np.zeros((1, 8), np.uint8))                                 # one filter output: assert at :356
    assert ei.value.status == -3


@pytest.mark.gpu
def test_gpu_fused_fuzz(fmd, oracle):
    """Random taps / decimation / shift / rates / call sizes, streaming (history, resampler phase and partial sums
    carried), against the oracle's composition.  FMD_FUZZ_CASES scales the number of cases."""
    import os
    n_cases = int(os.environ.get("FMD_FUZZ_CASES", "30"))
    rng = np.random.default_rng(int(os.environ.get("FMD_FUZZ_SEED", "777")))
    n_refused, kernels = 0, {}
    for case in range(n_cases):
        M = 2 * int(rng.choice([1, 2, 3, 4, 4, 4, 5, 8, 16, 25, 32]))      # (decimate 8 -- the register forms -- three times as often)
        T = int(rng.choice([1, 2, 3, 7, 16, 31, 64, 127, 128, 200, 513]))
        kind = int(rng.integers(0, 5))                                    # 12-bit taps, 8-bit taps (one tap digit), all ones
        taps = np.ones(T, np.int16) if kind == 0 else rng.integers(-127, 128, T).astype(np.int16) if kind <= 2 else rng.integers(-2047, 2048, T).astype(np.int16)
        shift = fmd.auto_shift(taps, 16384) + int(rng.integers(0, 7))     # both discriminator forms (|lp| <= 2048: f32)
        slow = int(rng.choice([8000, 32000, 44100, 48000]))
        fast = slow * int(rng.integers(1, 60)) + int(rng.integers(0, slow)) * int(rng.integers(0, 2))
        if M == 8 and rng.integers(0, 10) < 7:                            # most decimate-8 cases: inside the register forms' domain
            fast = slow * int(rng.integers(16, 60)) + int(rng.integers(0, slow)) * int(rng.integers(0, 2))
            shift = fmd.auto_shift(taps, 2048) + int(rng.integers(0, 3))
        nch = int(rng.integers(1, 5))
        try:
            fd = fmd.FirDemodBank(taps, M, fast, slow, nch, shift=shift)
        except fmd.FmdError as e:
            assert e.status == -6, (T, M, fast, slow, e)                     # outside the documented domain only
            n_refused += 1
            continue
        hs = [oracle.firdemod_new(taps, M, shift, fast, slow) for _ in range(nch)]
        first = 8 * ((T + 2 * M) // 4 + 2)
        for call in range(int(rng.integers(2, 6))):
            n = first if call == 0 else 8 * int(rng.integers(M, 2500))
            iq = rng.integers(0, 256, (nch, n), dtype=np.uint8)
            if rng.integers(0, 5) == 0:
                iq[:] = np.where(rng.integers(0, 2, (nch, n)) > 0, 255, 0)
            got = fd.demodulate_batch(iq)
            for c in range(nch):
                exp = oracle.firdemod(hs[c], iq[c])
                assert got[c].shape == exp.shape and np.array_equal(got[c], exp), (case, T, M, shift, fast, slow, call, c, n)
        for c in range(nch):
            assert state_tuple(fd.get_state(c).as_dict()) == state_tuple(oracle.firdemod_state(hs[c])), (case, c)
            oracle.lib.fmo_firdemod_free(hs[c])
        kn = fd.kernel_name().split("::")[-1].split("<")[0]
        kernels[kn] = kernels.get(kn, 0) + 1
        fd.close()
    print("fused fuzz: %d cases, %d outside the domain, kernels %s" % (n_cases, n_refused, kernels))
    assert n_refused * 2 <= n_cases, (n_refused, kernels)           # the fuzz must mostly RUN ...
    if n_cases >= 30:                                               # ... and reach the LDS-array kernel and the register forms
        assert "fmd_firdemod_kernel" in kernels and sum(v for k, v in kernels.items() if "reg" in k) >= n_cases // 8, kernels


@pytest.mark.gpu
@pytest.mark.parametrize("T,M,fast,slow", [(127, 8, 2500000, 48000), (33, 4, 250000, 48000), (64, 16, 1000000, 44100), (1, 2, 48000, 48000)])
def test_gpu_fused_checkpoint_resume(fmd, oracle, T, M, fast, slow):
    """The bank as a whole is the unit of checkpoint / resume (its channels advance together): a fresh bank resumed from
    the blob continues bit for bit as the original and as the oracle fed the uninterrupted stream; a blob from another
    filter, a truncated or a damaged one is refused (FMD_ERR_BAD_STATE = -7) and leaves the bank as it was."""
    rng = np.random.default_rng(T * 7 + M)
    taps = rng.integers(-2047, 2048, T).astype(np.int16)
    shift = fmd.auto_shift(taps, 2048)
    nch = 9
    a = fmd.FirDemodBank(taps, M, fast, slow, nch, shift=shift)
    hs = [oracle.firdemod_new(taps, M, shift, fast, slow) for _ in range(nch)]
    sizes = [8 * int(rng.integers(T // 4 + 2 * M + 4, 900)) for _ in range(5)]
    blocks = [rng.integers(0, 256, (nch, n), dtype=np.uint8) for n in sizes]
    for iq in blocks[:2]:
        got = a.demodulate_batch(iq)
        for c in range(nch):
            assert np.array_equal(got[c], oracle.firdemod(hs[c], iq[c])), c
    blob = a.checkpoint()
    assert len(blob) == 64 + nch * 32 + nch * 4 * ((T - 1 + ((T - 1) & 1)) // 2)   # header + FmdChanState + history words
    b = fmd.FirDemodBank(taps, M, fast, slow, nch, shift=shift)
    b.resume(blob)
    for c in range(nch):
        assert b.get_state(c).as_dict() == a.get_state(c).as_dict()
    for iq in blocks[2:]:
        ga, gb = a.demodulate_batch(iq), b.demodulate_batch(iq)
        for c in range(nch):
            exp = oracle.firdemod(hs[c], iq[c])
            assert np.array_equal(ga[c], exp) and np.array_equal(gb[c], exp), c
    for c in range(nch):
        assert state_tuple(b.get_state(c).as_dict()) == state_tuple(oracle.firdemod_state(hs[c]))
        oracle.lib.fmo_firdemod_free(hs[c])
    assert b.checkpoint() == a.checkpoint()

    # refusals: nothing of the bank changes
    before = b.checkpoint()
    other = fmd.FirDemodBank(np.where(taps > 0, taps - 1, taps + 1).astype(np.int16), M, fast, slow, nch, shift=shift)
    fewer = fmd.FirDemodBank(taps, M, fast, slow, nch - 1, shift=shift)
    bad_magic = b"\0" + blob[1:]

    def resealed(edit):
        """A hand-made blob: `edit` changes bytes, then the checksum (FNV-1a 64 of the header with that field zeroed + the
        payload, the last 8 header bytes) is recomputed -- so only the range checks stand between it and the kernels."""
        d = bytearray(blob)
        edit(d)
        d[56:64] = bytes(8)
        h = 0xCBF29CE484222325
        for x in d:
            h = ((h ^ x) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        d[56:64] = h.to_bytes(8, "little")
        return bytes(d)

    def poke_i32(off, v):
        return lambda d: d.__setitem__(slice(off, off + 4), int(v).to_bytes(4, "little", signed=True))

    ch0 = 64                                                           # FmdChanState of channel 0: prev_index, lpr_index_r, now_lpr,
    bound = -(-128 * int(np.abs(taps.astype(np.int64)).sum()) // (1 << shift))    # lp_now_re, lp_now_im, demod_pre_re, demod_pre_im, reserved
    group = 16384 * -(-(fast // math.gcd(fast, slow)) // (slow // math.gcd(fast, slow)))
    assert resealed(lambda d: None) == blob                            # the checksum restated here is the library's
    flipped = bytearray(blob); flipped[-1] ^= 0x40                     # one bit of the last channel's filter history
    flipped_state = bytearray(blob); flipped_state[ch0 + 20] ^= 1      # channel 0's demod_pre_re, checksum left alone
    refused = [(other, blob), (fewer, blob), (b, blob[:-4]), (b, blob + b"\0\0\0\0"), (b, blob[:20]), (b, bad_magic),
               (b, bytes(flipped)), (b, bytes(flipped_state)),
               (b, resealed(poke_i32(ch0 + 4, 1 + int.from_bytes(blob[ch0 + 4:ch0 + 8], "little")))),   # a channel off the bank's phase
               (b, resealed(poke_i32(ch0 + 20, bound + 1))), (b, resealed(poke_i32(ch0 + 24, -bound - 1))),   # |demod_pre| beyond the filter's range
               (b, resealed(poke_i32(ch0 + 32 * (nch - 1) + 8, group + 1))),                              # |now_lpr| beyond one audio group
               (b, resealed(poke_i32(ch0 + 28, 1))), (b, resealed(poke_i32(ch0 + 0, 1))),                  # reserved / prev_index must be 0
               (b, resealed(poke_i32(40, fast // math.gcd(fast, slow))))]                                  # header i0r >= fr
    for target, data in refused:
        with pytest.raises(fmd.FmdError) as ei:
            target.resume(data)
        assert ei.value.status == -7, ei.value
    # the extremes themselves are legal states
    b.resume(resealed(lambda d: (poke_i32(ch0 + 20, bound)(d), poke_i32(ch0 + 8, -group)(d))))
    st = b.get_state(0).as_dict()
    assert st["demod_pre"][0] == bound and st["now_lpr"] == -group
    b.resume(before)
    assert b.checkpoint() == before
    # a resumed-then-reset bank is a fresh one
    b.reset()
    c0 = fmd.FirDemodBank(taps, M, fast, slow, nch, shift=shift)
    assert b.checkpoint() == c0.checkpoint()
