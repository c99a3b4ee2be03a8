"""GPU parity: the HIP path, called through the C ABI (include/fmd.h), against the CPU oracle.

Bar: bit-exact s16 output and bit-exact Demod state (integer path).  The one f64 atan2 sample
per call (simple_fm.rs:359,370-374) is compared exactly too; see test_f64_sample_* for how
that is pinned.  Nothing here reads /root/reference.
"""
import ctypes as C
import math
import os

import numpy as np
import pytest

import oracle_lib
from conftest import run_in_exp_child

pytestmark = pytest.mark.gpu

CFG_REF = (6, 170000, 32000)      # optimal_settings(94.9 MHz, 170 kHz): simple_fm.rs:25-27,48
CFG_24 = (10, 240000, 32000)      # 2.4 Msps synthetic configuration (BASELINE.json configs[1..2])


def mkcfg(fmd, D, fast, slow):
    return fmd.DemodConfig(fast, fast, slow, D, max(1, (1 << 15) // (128 * D)))


def gpu_state(bank, ch=0):
    return bank.get_state(ch).as_dict()


def check_stream(fmd, oracle, D, fast, slow, blocks, n_channels=1, kt=None):
    """Feed the same blocks to GPU bank and oracle Demods; compare audio + state after every call."""
    cfg = mkcfg(fmd, D, fast, slow)
    bank = fmd.DemodBank(cfg, n_channels)
    if kt is not None:
        bank.set_tiling(kt)
    ocfg = oracle.config(D, fast, slow)
    obank = oracle.new_bank(ocfg, n_channels)
    for iq in blocks:
        got = bank.demodulate_batch(iq)
        exp, lens = oracle.demodulate_batch(obank, iq)
        for c in range(n_channels):
            assert got[c].size == lens[c], (c, got[c].size, lens[c])
            if not np.array_equal(got[c], exp[c, :lens[c]]):
                bad = np.nonzero(got[c] != exp[c, :lens[c]])[0]
                raise AssertionError("channel %d: %d mismatches, first at %d: gpu %d oracle %d" % (
                    c, bad.size, bad[0], got[c][bad[0]], exp[c, bad[0]]))
        for c in sorted(set([0, n_channels - 1, n_channels // 2])):
            assert gpu_state(bank, c) == oracle.state_of(obank[c]), c
    bank.close()


def test_library_and_device(fmd):
    assert fmd.device_count() >= 1, "no gfx950 device: the product has no CPU path"
    assert fmd.lib().fmd_version() >= 1
    import torch
    # the f64 sample is guarded against ocml / libm differences (test_gpu_f64_guard.py); the versions these results
    # were produced with are recorded anyway
    print("runtime: hip %s, torch %s, %s" % (getattr(torch.version, "hip", None), torch.__version__, torch.cuda.get_device_name(0)))


def test_synth_gpu_equals_numpy(fmd):
    import torch
    for (nch, nbytes, off) in [(3, 4096, 0), (70, 8 * 131, 12345678901)]:
        buf = torch.empty((nch, nbytes), dtype=torch.uint8, device="cuda")
        fmd.synth.fill_device(buf.data_ptr(), nch, nbytes, sample_offset=off)
        torch.cuda.synchronize()
        assert np.array_equal(buf.cpu().numpy(), fmd.synth.synth_iq(nch, nbytes, sample_offset=off))


def test_config1_reference_block_stream(fmd, oracle):
    """cfg-ref, 1 channel, 8 x DEFAULT_BUF_LENGTH blocks of the synthetic capture:
    boxcar phases 0/2/4, seven f64 call boundaries, a full-scale segment that wraps fast_atan2."""
    N = fmd.DEFAULT_BUF_LENGTH
    data = fmd.synth.synth_iq(1, 8 * N, seed=0x05D50001, amplitude=120)[0].copy()
    rng = np.random.default_rng(1)
    data[3 * N: 3 * N + 65536] = rng.integers(0, 256, 65536, dtype=np.uint8)           # full 0..255 coverage
    data[5 * N: 5 * N + 32768] = np.tile(np.array([255, 255, 0, 255, 0, 0, 255, 0], np.uint8), 4096)  # full scale DC
    check_stream(fmd, oracle, *CFG_REF, [data[i * N:(i + 1) * N][None, :] for i in range(8)])


def test_config2_one_channel_2p4msps(fmd, oracle):
    N = fmd.DEFAULT_BUF_LENGTH
    data = fmd.synth.synth_iq(1, 4 * N)[0]
    check_stream(fmd, oracle, *CFG_24, [data[i * N:(i + 1) * N][None, :] for i in range(4)])


def test_demod_single_stream_api(fmd, oracle):
    """Demod::new + demodulate through the single-stream entry point."""
    cfg = mkcfg(fmd, *CFG_REF)
    d = fmd.Demod(cfg)
    od = oracle.new(oracle.config(*CFG_REF))
    rng = np.random.default_rng(3)
    for _ in range(3):
        buf = rng.integers(0, 256, 8192, dtype=np.uint8)
        assert np.array_equal(d.demodulate(buf), oracle.demodulate(od, buf))
    assert gpu_state(d) == oracle.state_of(od)


@pytest.mark.parametrize("D,fast,slow", [CFG_REF, CFG_24, (7, 166666, 32000), (1, 48000, 48000), (5, 250000, 44100),
                                         (8, 128000, 32000), (3, 340000, 48000), (128, 8000, 8000), (21, 50000, 32000),
                                         (2, 1000000, 8000), (4, 60000, 60000), (10, 24000, 24000), (6, 32000, 32000)])
def test_configs_batched_random(fmd, oracle, D, fast, slow):
    rng = np.random.default_rng(D * 13 + 1)
    nch = 9
    blocks = []
    for i in range(4):
        n = int(rng.integers(2, 400)) * 8 + 16 * D
        if i % 2:
            blk = rng.integers(0, 256, (nch, n), dtype=np.uint8)
        else:
            blk = np.where(rng.integers(0, 2, (nch, n)) > 0, 255, 0).astype(np.uint8)   # full scale
        blocks.append(blk)
    check_stream(fmd, oracle, D, fast, slow, blocks, n_channels=nch)


@pytest.mark.parametrize("D,fast,slow", [(2, 500000, 32000), (4, 256000, 48000), CFG_REF, (7, 166666, 32000), CFG_24,
                                         (13, 208000, 32000), (16, 150000, 32000)])
def test_near_silence(fmd, oracle, D, fast, slow):
    """Bytes within +-2 of the centre: decimated samples are small, components and whole products are zero all the
    time -- (0, 0) samples, x == 0 with y != 0, y == 0 with either sign of x.  fast_atan2 (simple_fm.rs:383-405)
    branches on x >= 0 and y < 0, where zero is not negative; an f32 form that reads sign bits must not see a -0
    (found by the randomised test in one sample of 10^5 when the complex product moved to f32 multiplies)."""
    rng = np.random.default_rng(1000 + D)
    nch = 6
    blocks = []
    for i in range(3):
        n = int(rng.integers(200, 900)) * 8 + 16 * D
        blk = rng.integers(126, 131, (nch, n)).astype(np.uint8)
        if i == 1:
            blk[:, ::2] = 128                                  # I constant, Q moving: one component identically small
        if i == 2:
            blk[rng.integers(0, 2, (nch, n)) > 0] = 128        # sparse
        blocks.append(blk)
    check_stream(fmd, oracle, D, fast, slow, blocks, n_channels=nch)


def axis_pattern(re, im):
    """8 raw bytes whose rotated (rotate_90, :284-296) and centred (`- 127`, :258) samples are all (re, im), for re, im
    in {-127, 0, 128}: sample n of each 4 is (b0-127, b1-127), (128-b3, b2-127), (128-b4, 128-b5), (b7-127, 128-b6)."""
    p = lambda v: v + 127          # plain byte
    n = lambda v: 128 - v          # byte behind a `255 - x`
    return np.array([p(re), p(im), p(im), n(re), n(re), n(im), n(im), p(re)], np.uint8)


@pytest.mark.parametrize("D,fast,slow", [(4, 256000, 48000), CFG_REF, (7, 166666, 32000), CFG_24, (12, 192000, 32000),
                                         (16, 150000, 32000), (18, 180000, 30000)])
def test_axis_aligned_full_scale(fmd, oracle, D, fast, slow):
    """Decimated samples that sit exactly on an axis at full scale, (+-F, 0) and (0, +-F) in random order: the products are
    x = +-0 with |y| = F^2 >= 2^19 and y = +-0 with |x| = F^2 -- both signs of zero together with the i32 wrap of
    `(4096 * s) as i32` (:397,399).  An f32 discriminator that reads sign bits must canonicalise BOTH components: with
    x = -0 and y = 2^19 the two branches of fast_atan2 differ (16384 vs 8192) -- a = (0, -762), b = (-762, 0) at downsample 6."""
    rng = np.random.default_rng(3000 + D)
    nch = 5
    seg = 8 * D                                               # bytes: 4 whole windows, a multiple of the 8-byte rotation period
    vals = [(128, 0), (-127, 0), (0, 128), (0, -127), (0, 0)]
    blocks = []
    for i in range(3):
        nseg = int(rng.integers(20, 60))
        blk = np.empty((nch, nseg * seg), np.uint8)
        for c in range(nch):
            picks = rng.integers(0, len(vals) - (0 if c == 0 else 1), nseg)       # channel 0 also passes through (0, 0)
            blk[c] = np.concatenate([np.tile(axis_pattern(*vals[k]), seg // 8) for k in picks])
        blocks.append(blk)
    check_stream(fmd, oracle, D, fast, slow, blocks, n_channels=nch)


@pytest.mark.parametrize("D,fast,slow", [(2, 500000, 32000), (3, 400000, 48000), (4, 256000, 48000), (4, 200000, 32000), (5, 250000, 44100),
                                         CFG_REF, (8, 250000, 44100), CFG_24, (16, 150000, 32000)])
def test_diagonal_full_scale(fmd, oracle, D, fast, slow):
    """Decimated samples on the DIAGONALS at full scale, (+-F, +-F) in random order with the axes mixed in: a = b = (128 D, 128 D) gives
    x = 2 (128 D)^2, y = 0 -- at downsample 4 exactly 2^19, the ONE product for which `(4096 * s) as i32` (:397) wraps at that factor
    (s = +2^19 -> -2^31; every other s lies in [-520192, 2^19)), which the reference turns into 8192 where the angle is 0; from
    downsample 5 on the wrap is two-sided.  The kernels' single-point form of the wrap at downsample 4 (fmd_device.h) stands on this."""
    rng = np.random.default_rng(4000 + D)
    nch = 5
    seg = 8 * D
    vals = [(128, 128), (-127, -127), (128, -127), (-127, 128), (128, 0), (0, 128), (-127, 0), (0, -127), (0, 0)]
    blocks = []
    for i in range(3):
        nseg = int(rng.integers(30, 80))
        blk = np.empty((nch, nseg * seg), np.uint8)
        for c in range(nch):
            picks = rng.integers(0, 4 if c == 1 else len(vals), nseg)             # channel 1: diagonals only
            if c == 2:
                picks[:] = 0                                                      # channel 2: the saturated constant -- every product is the wrap point
            blk[c] = np.concatenate([np.tile(axis_pattern(*vals[k]), seg // 8) for k in picks])
        blocks.append(blk)
    check_stream(fmd, oracle, D, fast, slow, blocks, n_channels=nch)


@pytest.mark.parametrize("kt", [1, 2, 7, 64, 300])
def test_tiling_invariance(fmd, oracle, kt):
    rng = np.random.default_rng(kt)
    blocks = [rng.integers(0, 256, (5, 40000), dtype=np.uint8) for _ in range(2)]
    check_stream(fmd, oracle, *CFG_24, blocks, n_channels=5, kt=kt)
    check_stream(fmd, oracle, *CFG_REF, blocks, n_channels=5, kt=kt)


def test_ragged_and_tiny_calls(fmd, oracle):
    """Smallest legal calls (2 decimated samples), calls producing no audio, odd 8-byte multiples."""
    rng = np.random.default_rng(9)
    for D, fast, slow in [CFG_REF, CFG_24, (7, 166666, 32000)]:
        sizes = [4 * D + 8 - (4 * D) % 8, 8 * D, 24 * D + 8, 8, 8 * D * 3, 4096 + 8]
        sizes = [s for s in sizes if s % 8 == 0]
        cfg = mkcfg(fmd, D, fast, slow)
        bank = fmd.DemodBank(cfg, 2)
        obank = oracle.new_bank(oracle.config(D, fast, slow), 2)
        for n in sizes:
            iq = rng.integers(0, 256, (2, n), dtype=np.uint8)
            M = (int(obank[0].prev_index) + n // 2) // D
            if M < 2:
                with pytest.raises(fmd.FmdError) as ei:
                    bank.demodulate_batch(iq)
                assert ei.value.status == -3
                continue
            got = bank.demodulate_batch(iq)
            exp, lens = oracle.demodulate_batch(obank, iq)
            for c in range(2):
                assert np.array_equal(got[c], exp[c, :lens[c]])
            assert gpu_state(bank, 1) == oracle.state_of(obank[1])


def test_error_behaviour(fmd):
    cfg = mkcfg(fmd, *CFG_REF)
    d = fmd.Demod(cfg)
    with pytest.raises(fmd.FmdError) as ei:
        d.demodulate(np.zeros(12, np.uint8))            # reference: index panic (simple_fm.rs:286)
    assert ei.value.status == -2
    with pytest.raises(fmd.FmdError) as ei:
        d.demodulate(np.zeros(16, np.uint8))            # reference: assert (simple_fm.rs:356)
    assert ei.value.status == -3
    assert d.get_state().as_dict()["prev_index"] == 0    # failed calls leave the state untouched
    out = np.empty(1, np.int16)
    n = C.c_size_t()
    buf = np.zeros(8192, np.uint8)
    rc = fmd.lib().fmd_demod_demodulate(d._h, buf.ctypes.data, buf.size, out.ctypes.data, 1, C.byref(n))
    assert rc == -5
    for bad in [(6, 32000, 170000), (0, 170000, 32000), (6, 170000, 0), (513, 1950, 1950)]:                      # 129 ... 512: the generic kernel (test_downsample_129_to_512)
        with pytest.raises(fmd.FmdError):
            fmd.Demod(mkcfg(fmd, *bad) if bad[0] else fmd.DemodConfig(bad[1], bad[1], bad[2], 0, 1))
    with pytest.raises(fmd.FmdError):
        fmd.DemodBank(cfg, 0)


def test_state_checkpoint_resume(fmd, oracle):
    """get_state / set_state == the reference's resumable Demod fields (simple_fm.rs:232-239)."""
    rng = np.random.default_rng(21)
    D, fast, slow = CFG_REF
    a = rng.integers(0, 256, 50008, dtype=np.uint8)
    b = rng.integers(0, 256, 30000, dtype=np.uint8)
    od = oracle.new(oracle.config(D, fast, slow))
    oracle.demodulate(od, a)
    exp_b = oracle.demodulate(od, b)
    g1 = fmd.Demod(mkcfg(fmd, D, fast, slow))
    g1.demodulate(a)
    st = g1.get_state()
    g2 = fmd.DemodBank(mkcfg(fmd, D, fast, slow), 3)
    g2.set_state(1, st)
    got = g2.demodulate_batch(np.stack([b, b, b]))
    assert np.array_equal(got[1], exp_b)
    exp0 = oracle.demodulate(oracle.new(oracle.config(D, fast, slow)), b)   # channel 0 started from the zero state
    assert np.array_equal(got[0], exp0) and np.array_equal(got[2], exp0)
    assert g2.get_state(1).as_dict() == oracle.state_of(od)
    bad = fmd.DemodState(prev_index=D)                   # unreachable phase
    with pytest.raises(fmd.FmdError) as ei:
        g2.set_state(0, bad)
    assert ei.value.status == -7


def test_f64_sample_special_directions_and_random(fmd, oracle):
    """The per-call f64 sample (simple_fm.rs:359,370-374) with a NON-ZERO predecessor, which no
    reference test covers.  fast == slow makes every discriminator sample an output, so the f64
    sample is observable directly.  The state is injected through set_state on both sides."""
    D = 4
    cfg = mkcfg(fmd, D, 48000, 48000)
    ocfg = oracle.config(D, 48000, 48000)
    rng = np.random.default_rng(77)
    pres, lps = [], []
    for v in [(1, 0), (0, 1), (-1, 0), (0, -1), (1, 1), (-1, 1), (1, -1), (-1, -1), (3, 4), (200, -311)]:
        for s in ([1, 7, 100] if max(abs(v[0]), abs(v[1])) == 1 else ([1, 7, 50] if v == (3, 4) else [1])):
            pres.append((1, 0)); lps.append((v[0] * s, v[1] * s))            # c = lp0 exactly
            pres.append((v[0] * s, v[1] * s)); lps.append((v[0] * s, v[1] * s))  # c real positive
            pres.append((v[0] * s, v[1] * s)); lps.append((-v[1] * s, v[0] * s))  # c = +j |.|^2
    for _ in range(600):
        pres.append(tuple(int(x) for x in rng.integers(-512, 513, 2)))
        lps.append(tuple(int(x) for x in rng.integers(-384, 385, 2)))
    n = len(pres)
    bank = fmd.DemodBank(cfg, n)
    obank = oracle.new_bank(ocfg, n)
    for c in range(n):
        # phase D-1 with lp_now = lps: the first input sample (bytes 127,127 -> (0,0)) completes lp[0] = lps
        st = fmd.DemodState(prev_index=D - 1, now_lpr=0, prev_lpr_index=0, lp_now_re=lps[c][0], lp_now_im=lps[c][1],
                            demod_pre_re=pres[c][0], demod_pre_im=pres[c][1])
        bank.set_state(c, st)
        obank[c].prev_index = D - 1
        obank[c].lp_now.re, obank[c].lp_now.im = lps[c]
        obank[c].demod_pre.re, obank[c].demod_pre.im = pres[c]
    iq = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    iq[:, 0:2] = 127
    got = bank.demodulate_batch(iq)
    exp, lens = oracle.demodulate_batch(obank, iq)
    bad = [(c, int(got[c][0]), int(exp[c, 0])) for c in range(n) if got[c][0] != exp[c, 0]]
    assert not bad, bad[:10]
    for c in range(n):
        assert np.array_equal(got[c], exp[c, :lens[c]])


def test_device_entry_point_and_out_len(fmd, oracle):
    """fmd_demod_demodulate_device on torch-owned HBM buffers + the device/host out_len."""
    import torch
    nch, N = 33, 65536
    cfg = mkcfg(fmd, *CFG_24)
    bank = fmd.DemodBank(cfg, nch)
    obank = oracle.new_bank(oracle.config(*CFG_24), nch)
    iq = torch.empty((nch, N), dtype=torch.uint8, device="cuda")
    cap = bank.out_cap(N)
    out = torch.zeros((nch, cap), dtype=torch.int16, device="cuda")
    lens = torch.zeros(nch, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for call in range(3):
        fmd.synth.fill_device(iq.data_ptr(), nch, N, sample_offset=call * (N // 2), stream=stream)
        bank.demodulate_device(iq.data_ptr(), N, out.data_ptr(), cap, lens.data_ptr(), stream)
        torch.cuda.synchronize()
        exp, elens = oracle.demodulate_batch(obank, iq.cpu().numpy())
        assert np.array_equal(lens.cpu().numpy().astype(np.uint32), elens)
        assert np.array_equal(bank.last_out_len().astype(np.uint32), elens)
        o = out.cpu().numpy()
        for c in range(nch):
            assert np.array_equal(o[c, :elens[c]], exp[c, :elens[c]])


def test_full_size_config3_exact(fmd, oracle):
    """BASELINE configs[2]: 4096 channels x 262144 B at cfg-2.4 on one GPU, two consecutive calls,
    compared EXACTLY against the multi-threaded oracle (1 GiB per call), plus the size-independent
    property that identical channels give identical audio."""
    import torch
    nch, N = 4096, fmd.DEFAULT_BUF_LENGTH
    cfg = mkcfg(fmd, *CFG_24)
    bank = fmd.DemodBank(cfg, nch)
    obank = oracle.new_bank(oracle.config(*CFG_24), nch)
    iq = torch.empty((nch, N), dtype=torch.uint8, device="cuda")
    cap = bank.out_cap(N)
    out = torch.zeros((nch, cap), dtype=torch.int16, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for call in range(2):
        fmd.synth.fill_device(iq.data_ptr(), nch, N, sample_offset=call * (N // 2), stream=stream)
        iq[7] = iq[3]                                   # two identical channels
        bank.demodulate_device(iq.data_ptr(), N, out.data_ptr(), cap, None, stream)
        torch.cuda.synchronize()
        host = iq.cpu().numpy()
        exp, elens = oracle.demodulate_batch(obank, host)
        o = out.cpu().numpy()
        assert np.array_equal(bank.last_out_len().astype(np.uint32), elens)
        K = int(elens.max())
        mask = np.arange(K)[None, :] < elens[:, None]
        assert np.array_equal(np.where(mask, o[:, :K], 0), np.where(mask, exp[:, :K], 0))
        assert np.array_equal(o[7, :elens[7]], o[3, :elens[3]])
    assert gpu_state(bank, 4095) == oracle.state_of(obank[4095])


def test_config5_shape_on_one_gpu(fmd, oracle):
    """BASELINE configs[4] (32768 channels over 8 GPUs, 4096 each) as far as ONE GPU can show it: for every rank r of the
    8-GPU job, the rank's channel range, its input seeded exactly as bench.py seeds it (base seed + first global channel
    id) and the full 4096-channel x 262144 B launch; >= 64 channels per rank (first, last, strided) must equal the oracle
    fed the stream that global channel has in the 8-GPU job, and different ranks must produce different audio (one Demod per stream,
    simple_fm.rs:137: rank r's channels are not rank 0's)."""
    import torch
    world, per = 8, 4096
    N = fmd.DEFAULT_BUF_LENGTH
    base_seed = fmd.synth.DEFAULTS["seed"]
    cfg = mkcfg(fmd, *CFG_24)
    iq = torch.empty((per, N), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    first_audio = []
    for r in range(world):
        lo, hi = fmd.shard.channel_range(world * per, world, r)
        assert (lo, hi) == (r * per, (r + 1) * per)
        fmd.synth.fill_device(iq.data_ptr(), per, N, sample_offset=0, stream=stream, seed=base_seed + lo)   # bench.py's seeding
        bank = fmd.DemodBank(cfg, per)
        cap = bank.out_cap(N)
        out = torch.zeros((per, cap), dtype=torch.int16, device="cuda")
        bank.demodulate_device(iq.data_ptr(), N, out.data_ptr(), cap, None, stream)
        bank.check()
        lens = bank.last_out_len()
        picks = sorted(set([0, 1, per - 2, per - 1] + list(range(0, per, 67))))
        assert len(picks) >= 64
        got = out[picks].cpu().numpy()
        # the oracle sees the stream global channel lo + c has in the 8-GPU job (seed = base + lo, row c -- bench.py's
        # seeding), generated independently on the host by the numpy generator
        host = np.concatenate([fmd.synth.synth_iq(1, N, seed=base_seed + lo, first_channel=c) for c in picks])
        assert np.array_equal(host, iq[picks].cpu().numpy())                 # device generator == host generator
        obank = oracle.new_bank(oracle.config(*CFG_24), len(picks))
        exp, elens = oracle.demodulate_batch(obank, host)
        for j, c in enumerate(picks):
            assert lens[c] == elens[j]
            assert np.array_equal(got[j, :elens[j]], exp[j, :elens[j]]), (r, c)
        assert gpu_state(bank, per - 1) == oracle.state_of(obank[len(picks) - 1])
        first_audio.append(got[0, :elens[0]].copy())
        bank.close()
        del out
    for r in range(1, world):
        assert not np.array_equal(first_audio[r], first_audio[0]), r


@pytest.mark.parametrize("D,fast,slow", [(4, 256000, 48000), (4, 192000, 48000), (2, 500000, 32000)])
def test_streaming_kernel_long_calls(fmd, oracle, D, fast, slow):
    """Downsample 2 / 4 with >= 8 channels (the register-streaming kernel) and calls of several MiB per channel: more tiles
    than the per-tile table holds.  192 k -> 48 k has tiles that repeat exactly (closed-form geometry: the streaming kernel
    keeps running), 256 k -> 48 k and 500 k -> 32 k do not (the library plans the call again for the LDS-DMA kernel) -- same
    audio and state either way, then a reference-sized call continues the stream on the streaming kernel again."""
    rng = np.random.default_rng(D + fast)
    nch = 9
    blocks = [rng.integers(0, 256, (nch, 3 << 20), dtype=np.uint8),
              fmd.synth.synth_iq(nch, fmd.DEFAULT_BUF_LENGTH, seed=7, amplitude=110),
              np.where(rng.integers(0, 2, (nch, 1 << 20)) > 0, 255, 0).astype(np.uint8)]
    check_stream(fmd, oracle, D, fast, slow, blocks, n_channels=nch)


def test_last_kernel_and_tiling_report_what_ran(fmd):
    """fmd_demod_last_kernel / fmd_demod_tiling describe the launch that actually happened (the tiling used to
    describe the streaming kernel whenever it COULD run, also when the call had fallen back to the LDS-DMA kernel)."""
    rng = np.random.default_rng(11)
    bank = fmd.DemodBank(mkcfg(fmd, 4, 256000, 48000), 9)
    assert bank.last_kernel() == ""
    planned = bank.tiling()                                    # before any launch: what whole read_sync buffers will run
    bank.demodulate_batch(rng.integers(0, 256, (9, 3 << 20), dtype=np.uint8))     # > 32 tiles, tiles do not repeat: LDS-DMA kernel, general prologue
    assert bank.last_kernel() == "fmd_tk::fmd_demod_tile_kernel<2, 0>"
    lds_tiling = bank.tiling()
    assert lds_tiling["lds_bytes"] > 8192 and lds_tiling["audio_per_tile"] < planned["audio_per_tile"]
    bank.demodulate_batch(rng.integers(0, 256, (9, fmd.DEFAULT_BUF_LENGTH), dtype=np.uint8))
    assert bank.last_kernel() == "fmd_tk::fmd_demod_stream_kernel<2, 2>" and bank.tiling() == planned
    bank.close()
    one = fmd.DemodBank(mkcfg(fmd, *CFG_24), 1)                 # < 8 channels: no XCD-aware grid, general prologue
    one.demodulate_batch(rng.integers(0, 256, (1, fmd.DEFAULT_BUF_LENGTH), dtype=np.uint8))
    assert one.last_kernel() == "fmd_tk::fmd_demod_tile_kernel<5, 0>"
    one.close()
    wide = fmd.DemodBank(mkcfg(fmd, 200, 5000, 5000), 2)
    wide.demodulate_batch(rng.integers(0, 256, (2, fmd.DEFAULT_BUF_LENGTH), dtype=np.uint8))
    assert wide.last_kernel() == "fmd_demod_generic_kernel<true>"
    wide.close()


def test_streaming_kernel_falls_back_bit_exactly(fmd, oracle):
    """The same two calls as above against the oracle: a 3 MiB call (more tiles than the table holds: the streaming kernel
    declines and the LDS-DMA kernel with the general prologue runs), then a read_sync buffer through the streaming kernel
    on the state the fallback left -- audio and state of every channel after each."""
    rng = np.random.default_rng(12)
    blocks = [rng.integers(0, 256, (9, 3 << 20), dtype=np.uint8), rng.integers(0, 256, (9, fmd.DEFAULT_BUF_LENGTH), dtype=np.uint8),
              fmd.synth.synth_iq(9, fmd.DEFAULT_BUF_LENGTH, sample_offset=77, amplitude=120)]
    check_stream(fmd, oracle, 4, 256000, 48000, blocks, n_channels=9)


@pytest.mark.parametrize("fast,slow", [(48000, 48000), (250000, 48000), (1024000, 32000), (96000, 44100)])
def test_downsample_1_adjacent_samples(fmd, oracle, fast, slow):
    """Downsample 1 has a round form of its own (one dword = two samples per lane, waves starting at even and at odd samples,
    per-lane rotation phases): random, full-scale square, silent and synthetic FM calls of many sizes, one channel and nine."""
    rng = np.random.default_rng(fast // 1000 + 7)
    for nch in (1, 9):
        blocks = [rng.integers(0, 256, (nch, 8 * int(rng.integers(3, 5000))), dtype=np.uint8) for _ in range(5)]
        blocks.append(np.where(rng.integers(0, 2, (nch, 8 * 777)) > 0, 255, 0).astype(np.uint8))
        blocks.append(np.full((nch, 8 * 300), 127, np.uint8))
        blocks.append(fmd.synth.synth_iq(nch, fmd.DEFAULT_BUF_LENGTH, amplitude=120))
        check_stream(fmd, oracle, 1, fast, slow, blocks, n_channels=nch)


def test_large_single_channel_call(fmd, oracle):
    """Config 2 throughput shape: one channel, 16 MiB in one call (time-tiled inside the channel)."""
    N = 16 << 20
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, N, dtype=np.uint8)
    check_stream(fmd, oracle, *CFG_24, [data[None, :]])


def test_generic_kernel_forced(fmd, oracle, request):
    """The fallback kernel (in-kernel index divisions; what the library itself takes for > 16 phase classes -- see
    test_phase_classes[23] -- and extreme rate ratios) stays bit-exact on ordinary streams too.  Forcing it is a knob of
    the -DFMD_EXPERIMENT build: the case re-runs itself there."""
    if run_in_exp_child(request, {"FMD_FORCE_GENERIC": "1"}):
        return
    rng = np.random.default_rng(31)
    blocks = [rng.integers(0, 256, (6, 65536), dtype=np.uint8) for _ in range(3)]
    check_stream(fmd, oracle, *CFG_24, blocks, n_channels=6)
    check_stream(fmd, oracle, *CFG_REF, blocks, n_channels=6)


@pytest.mark.parametrize("nclasses", [2, 4, 7, 16, 23])
def test_phase_classes(fmd, oracle, nclasses):
    """Channels with different call-start phases in one bank: <= 16 classes run the tile kernel with a
    per-channel class table, more fall back to the generic kernel.  Phases are desynchronised the way a
    caller could: by checkpointing a Demod that has consumed a different amount of input."""
    D, fast, slow = CFG_REF
    nch = 48
    rng = np.random.default_rng(nclasses)
    cfg = mkcfg(fmd, D, fast, slow)
    bank = fmd.DemodBank(cfg, nch)
    obank = oracle.new_bank(oracle.config(D, fast, slow), nch)
    for c in range(nch):
        k = c % nclasses
        if k == 0:
            continue
        pre = rng.integers(0, 256, 8 * (40 + 13 * k), dtype=np.uint8)    # different lengths -> different phases
        oracle.demodulate(obank[c], pre)
        s = oracle.state_of(obank[c])
        bank.set_state(c, fmd.DemodState(prev_index=s["prev_index"], now_lpr=s["now_lpr"],
                                         prev_lpr_index=s["prev_lpr_index"], lp_now_re=s["lp_now"][0],
                                         lp_now_im=s["lp_now"][1], demod_pre_re=s["demod_pre"][0],
                                         demod_pre_im=s["demod_pre"][1]))
    for _ in range(3):
        iq = rng.integers(0, 256, (nch, 30008), dtype=np.uint8)
        got = bank.demodulate_batch(iq)
        exp, lens = oracle.demodulate_batch(obank, iq)
        for c in range(nch):
            assert got[c].size == lens[c]
            assert np.array_equal(got[c], exp[c, :lens[c]]), c
    for c in range(nch):
        assert gpu_state(bank, c) == oracle.state_of(obank[c])


@pytest.mark.parametrize("force_generic", [False, True])
def test_more_than_65535_channels(fmd, oracle, request, force_generic):
    """Maximum-size edge: 70 001 channels in one bank (beyond one grid dimension), two small calls; a strided
    sample of channels incl. both sides of the 65535 boundary and the last one is compared with the oracle, and
    every channel that was given channel 5's input must reproduce channel 5's audio."""
    import torch
    if force_generic and run_in_exp_child(request, {"FMD_FORCE_GENERIC": "1"}):
        return
    nch, N = 70001, 4096
    cfg = mkcfg(fmd, *CFG_24)
    bank = fmd.DemodBank(cfg, nch)
    cap = bank.out_cap(N)
    iq = torch.empty((nch, N), dtype=torch.uint8, device="cuda")
    out = torch.zeros((nch, cap), dtype=torch.int16, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    picks = sorted(set(list(range(0, nch, 997)) + [65534, 65535, 65536, 65537, nch - 1]))
    obank = oracle.new_bank(oracle.config(*CFG_24), len(picks))
    for call in range(2):
        fmd.synth.fill_device(iq.data_ptr(), nch, N, sample_offset=call * (N // 2), stream=stream)
        iq[60000:] = iq[5]                              # 10 001 copies of channel 5 across the boundary
        bank.demodulate_device(iq.data_ptr(), N, out.data_ptr(), cap, None, stream)
        torch.cuda.synchronize()
        lens = bank.last_out_len()
        exp, elens = oracle.demodulate_batch(obank, iq[picks].cpu().numpy())
        o = out.cpu().numpy()
        for j, c in enumerate(picks):
            assert lens[c] == elens[j]
            assert np.array_equal(o[c, :elens[j]], exp[j, :elens[j]]), c
        assert (lens[60000:] == lens[5]).all()
        assert (o[60000:, :lens[5]] == o[5, :lens[5]][None, :]).all()
    assert gpu_state(bank, nch - 1) == oracle.state_of(obank[len(picks) - 1])
    bank.close()


def test_pinned_host_buffers(fmd, oracle):
    """fmd_host_alloc / fmd_host_free: page-locked read buffers through the host entry point give the same audio."""
    nch, N = 5, 32768
    cfg = mkcfg(fmd, *CFG_REF)
    bank = fmd.DemodBank(cfg, nch)
    obank = oracle.new_bank(oracle.config(*CFG_REF), nch)
    cap = bank.out_cap(N)
    pin_in, pin_out = fmd.PinnedBuffer((nch, N), np.uint8), fmd.PinnedBuffer((nch, cap), np.int16)
    rng = np.random.default_rng(77)
    for _ in range(3):
        pin_in.array[:] = rng.integers(0, 256, (nch, N), dtype=np.uint8)
        lens = bank.demodulate_batch_into(pin_in.array, pin_out.array)
        exp, elens = oracle.demodulate_batch(obank, pin_in.array)
        assert np.array_equal(lens.astype(np.uint32), elens)
        for c in range(nch):
            assert np.array_equal(pin_out.array[c, :elens[c]], exp[c, :elens[c]])
    pin_in.close(); pin_out.close()
    p = C.c_void_p()
    assert fmd.lib().fmd_host_alloc(0, C.byref(p)) == -1 and fmd.lib().fmd_host_free(None) == 0


@pytest.mark.parametrize("D,fast,slow", [(64, 15000, 8000), (81, 12500, 8000), (127, 8000, 8000), (128, 8000, 4000)])
def test_large_downsample_full_scale(fmd, oracle, D, fast, slow):
    """Narrow-band settings (optimal_settings(f, 12500) gives downsample 81): boxcar sums reach +-128*D and the
    discriminator's |x| + |y| approaches 2^30; rotated full-scale DC of either sign, abrupt sign flips (largest
    products of consecutive decimated samples) and random data, over ragged calls."""
    rng = np.random.default_rng(D)
    pos = np.array([255, 255, 0, 255, 0, 0, 255, 0], np.uint8)        # rotate_90 + centre -> (+128, +128) every sample
    neg = 255 - pos                                                   # -> (-127, -127)
    nch = 3
    blocks = []
    for i in range(4):
        n = 8 * int(rng.integers(6 * D, 14 * D))
        blk = np.empty((nch, n), np.uint8)
        for c in range(nch):
            segs, left = [], n
            while left > 0:
                m = min(left, 8 * int(rng.integers(1, 3 * D)))
                kind = int(rng.integers(0, 3))
                segs.append(np.tile(pos if kind == 0 else neg, m // 8) if kind < 2 else rng.integers(0, 256, m, dtype=np.uint8))
                left -= m
            blk[c] = np.concatenate(segs)
        blocks.append(blk)
    check_stream(fmd, oracle, D, fast, slow, blocks, n_channels=nch)


@pytest.mark.parametrize("D,fast,slow,kind", [(129, 7752, 3876, "full"), (200, 5000, 5000, "full"), (255, 3922, 1000, "full"),
                                              (304, 3290, 3290, "full"), (400, 2500, 1250, "full"), (512, 1957, 1957, "random"), (512, 1957, 1957, "full"),
                                              (512, 1957, 600, "dc")])
def test_downsample_129_to_512(fmd, oracle, D, fast, slow, kind):
    """optimal_settings(f, rate) gives downsample 1_000_000 / rate + 1 (simple_fm.rs:190): 129 ... 512 for rates from 7812
    down to 1957 Hz.  Those run the generic kernel with i32 decimated samples, and the reference's own i32 arithmetic
    starts to wrap there -- `x + yabs` in fast_atan2 beyond 128, the complex product beyond 255 -- which must be
    reproduced as the wrapping operations of a release build (the oracle: -fwrapv).  "full": rotated full-scale DC of
    either sign with abrupt flips plus random data (the largest products); "random" / "dc": the same ingredients one at a
    time at the factors where some full-scale mixtures make the REFERENCE panic (zero divisor after the wrap, downsample
    >= 305): the oracle counts such samples and the test asserts there were none."""
    rng = np.random.default_rng(D)
    pos = np.array([255, 255, 0, 255, 0, 0, 255, 0], np.uint8)        # rotate_90 + centre -> (+128, +128) every sample
    neg = 255 - pos                                                   # -> (-127, -127)
    nch = 3
    panics0 = oracle.lib.fmo_would_panic()
    blocks = []
    for i in range(3):
        n = 8 * int(rng.integers(5 * D, 9 * D))
        blk = np.empty((nch, n), np.uint8)
        for c in range(nch):
            segs, left = [], n
            while left > 0:
                m = min(left, 8 * int(rng.integers(1, 2 * D)))
                k = int(rng.integers(0, 3)) if kind == "full" else (2 if kind == "random" else int(rng.integers(0, 2)))
                segs.append(np.tile(pos if k == 0 else neg, m // 8) if k < 2 else rng.integers(0, 256, m, dtype=np.uint8))
                left -= m
            blk[c] = np.concatenate(segs)
        blocks.append(blk)
    check_stream(fmd, oracle, D, fast, slow, blocks, n_channels=nch)
    assert oracle.lib.fmo_would_panic() == panics0                    # the reference itself is defined on these inputs
    if D == 512 and kind == "full":
        # a state the handle itself produced must be restorable -- beyond downsample 128 `pcm as i16` of the wrapped
        # arithmetic (simple_fm.rs:362) can lie anywhere in +-32768, so the carried partial sum exceeds the 16384-per-sample bound
        # that set_state applies up to 128.  Full scale, checkpoint after every call, resume in a fresh bank, same audio.
        cfg = mkcfg(fmd, D, fast, slow)
        a, b = fmd.DemodBank(cfg, nch), fmd.DemodBank(cfg, nch)
        big = 0
        for blk in blocks:
            want = a.demodulate_batch(blk)
            got = b.demodulate_batch(blk)
            assert all(np.array_equal(want[c], got[c]) for c in range(nch))
            fresh = fmd.DemodBank(cfg, nch)
            for c in range(nch):
                st = a.get_state(c)
                big = max(big, abs(st.as_dict()["now_lpr"]))
                fresh.set_state(c, st)                                 # must not be FMD_ERR_BAD_STATE
                assert fresh.get_state(c).as_dict() == st.as_dict()
            b.close(); b = fresh
        a.close(); b.close()
    if D == 512:
        with pytest.raises(fmd.FmdError) as ei:
            fmd.DemodBank(mkcfg(fmd, 513, fast, slow), 1)
        assert ei.value.status == -6                                   # FMD_ERR_UNSUPPORTED beyond 512


@pytest.mark.parametrize("D,fast,slow,block", [(6, 170000, 32000, 262144), (10, 240000, 32000, 262144), (10, 240000, 32000, 4096),
                                               (7, 166666, 32000, 30008), (5, 250000, 44100, 1000), (16, 150000, 32000, 8192),
                                               (2, 500000, 32000, 64)])
def test_several_reference_calls_per_launch(fmd, oracle, D, fast, slow, block):
    """fmd_demod_set_block_len: one launch over B blocks == the oracle (the reference) fed those B blocks one by
    one -- same audio (incl. the f64 sample at every block start, simple_fm.rs:359) and same state; then a second
    launch continues the stream, and switching the mode off again gives single-call semantics back."""
    rng = np.random.default_rng(D + block)
    nch = 3
    bank = fmd.DemodBank(mkcfg(fmd, D, fast, slow), nch)
    bank.set_block_len(block)
    obank = oracle.new_bank(oracle.config(D, fast, slow), nch)
    for B in (5, 3):
        iq = rng.integers(0, 256, (nch, B * block), dtype=np.uint8)
        if B == 3:
            iq[:, :block] = np.where(rng.integers(0, 2, (nch, block)) > 0, 255, 0)       # a full-scale block
        got = bank.demodulate_batch(iq)
        for c in range(nch):
            parts = [oracle.demodulate(obank[c], iq[c, b * block:(b + 1) * block]) for b in range(B)]
            exp = np.concatenate(parts)
            assert got[c].size == exp.size and np.array_equal(got[c], exp), (B, c)
        for c in range(nch):
            assert gpu_state(bank, c) == oracle.state_of(obank[c])
    with pytest.raises(fmd.FmdError) as ei:
        bank.demodulate_batch(np.zeros((nch, block + 8), np.uint8))                     # not a multiple of the block
    assert ei.value.status == -2
    bank.set_block_len(0)
    iq = rng.integers(0, 256, (nch, 2 * block + 8 * D * 4), dtype=np.uint8)
    got = bank.demodulate_batch(iq)
    for c in range(nch):
        assert np.array_equal(got[c], oracle.demodulate(obank[c], iq[c]))
    with pytest.raises(fmd.FmdError):
        bank.set_block_len(12)
    if D >= 4:
        with pytest.raises(fmd.FmdError) as ei:
            bank.set_block_len(8 * (D // 4))                                            # < 2 decimated samples per block
        assert ei.value.status == -3
