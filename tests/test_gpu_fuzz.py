"""Randomised GPU parity: random rates / downsample / tilings / call sizes / desynchronised phases, HIP path (through
the C ABI) against the oracle, audio and state after every call.  FMD_FUZZ_CASES scales the number of cases
(default 40; the development runs used 600)."""
import math
import os

import numpy as np
import pytest

from test_gpu_parity import check_stream, gpu_state, mkcfg

pytestmark = pytest.mark.gpu

RATES = [8000, 11025, 12500, 16000, 22050, 24000, 32000, 44100, 48000]


def random_case(rng):
    D = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 24, 25, 27, 28, 30, 31, 32, 33, 50, 64,
                       81, 128]))      # every factor with a kernel of its own up to 18, a sample of the others, the catch-all
    slow = int(rng.choice(RATES))
    fast = int(slow * rng.uniform(1.0, 9.0)) if rng.random() < 0.7 else int(rng.choice([170000, 240000, 250000, 166666, 1000000 // D + 1]))
    fast = max(fast, slow)
    g = math.gcd(fast, slow)
    if fast // g > 1 << 20:
        fast = slow * int(rng.integers(1, 8))
    return D, fast, slow


def test_fuzz_banks(fmd, oracle):
    n_cases = int(os.environ.get("FMD_FUZZ_CASES", "40"))
    rng = np.random.default_rng(int(os.environ.get("FMD_FUZZ_SEED", "20260101")))
    done = 0
    while done < n_cases:
        D, fast, slow = random_case(rng)
        nch = int(rng.integers(1, 7))
        kt = None if rng.random() < 0.5 else int(rng.integers(1, 400))
        blocks = []
        for _ in range(int(rng.integers(1, 5))):
            n = 8 * int(rng.integers(max(1, D // 2), 40 * D + 600))
            mode = rng.random()
            if mode < 0.6:
                blk = rng.integers(0, 256, (nch, n), dtype=np.uint8)
            elif mode < 0.8:
                blk = np.where(rng.integers(0, 2, (nch, n)) > 0, 255, 0).astype(np.uint8)
            else:
                blk = fmd.synth.synth_iq(nch, n, seed=int(rng.integers(1, 1 << 30)), amplitude=int(rng.integers(1, 121)))
            blocks.append(blk)
        try:
            check_stream(fmd, oracle, D, fast, slow, blocks, n_channels=nch, kt=kt)
        except fmd.FmdError as e:                        # a legal refusal (too short, LDS, ranges) -- never a wrong answer
            assert e.status in (-3, -5, -6), (D, fast, slow, nch, kt, e)
            continue
        except AssertionError as e:
            raise AssertionError("case D=%d fast=%d slow=%d nch=%d kt=%s sizes=%s: %s" % (
                D, fast, slow, nch, kt, [b.shape[1] for b in blocks], e))
        done += 1


def test_fuzz_desynchronised_phases(fmd, oracle):
    """Channels moved to random reachable states (1..6 phase classes: tile kernel with a class table, then the
    generic kernel), random rates."""
    n_cases = max(4, int(os.environ.get("FMD_FUZZ_CASES", "40")) // 5)
    rng = np.random.default_rng(77)
    for _ in range(n_cases):
        D, fast, slow = random_case(rng)
        nch, ncls = 8, int(rng.integers(1, 7))
        cfg = mkcfg(fmd, D, fast, slow)
        try:
            bank = fmd.DemodBank(cfg, nch)
        except fmd.FmdError as e:
            assert e.status == -6
            continue
        obank = oracle.new_bank(oracle.config(D, fast, slow), nch)
        for c in range(nch):
            k = c % ncls
            if k:
                pre = rng.integers(0, 256, 8 * (2 * D + 5 * k + int(rng.integers(0, 9))), dtype=np.uint8)
                oracle.demodulate(obank[c], pre)
                s = oracle.state_of(obank[c])
                bank.set_state(c, fmd.DemodState(prev_index=s["prev_index"], now_lpr=s["now_lpr"],
                                                 prev_lpr_index=s["prev_lpr_index"], lp_now_re=s["lp_now"][0],
                                                 lp_now_im=s["lp_now"][1], demod_pre_re=s["demod_pre"][0],
                                                 demod_pre_im=s["demod_pre"][1]))
        for _ in range(2):
            n = 8 * int(rng.integers(2 * D + 2, 30 * D + 500))
            iq = rng.integers(0, 256, (nch, n), dtype=np.uint8)
            got = bank.demodulate_batch(iq)
            exp, lens = oracle.demodulate_batch(obank, iq)
            for c in range(nch):
                assert got[c].size == lens[c] and np.array_equal(got[c], exp[c, :lens[c]]), (D, fast, slow, ncls, c)
        for c in range(nch):
            assert gpu_state(bank, c) == oracle.state_of(obank[c]), (D, fast, slow, ncls, c)
        bank.close()


def test_fuzz_several_reference_calls_per_launch(fmd, oracle):
    """Random rates / block lengths / block counts with fmd_demod_set_block_len: one launch == the oracle fed block
    by block."""
    n_cases = max(8, int(os.environ.get("FMD_FUZZ_CASES", "40")) // 2)
    rng = np.random.default_rng(int(os.environ.get("FMD_FUZZ_SEED", "20260101")) + 5)
    done = 0
    while done < n_cases:
        D, fast, slow = random_case(rng)
        nch = int(rng.integers(1, 4))
        block = 8 * int(rng.integers(max(1, (4 * D + 7) // 8), 12 * D + 300))
        try:
            bank = fmd.DemodBank(mkcfg(fmd, D, fast, slow), nch)
            bank.set_block_len(block)
        except fmd.FmdError as e:
            assert e.status in (-3, -6), (D, fast, slow, block, e)
            continue
        if rng.random() < 0.5:
            try:
                bank.set_tiling(int(rng.integers(1, 300)))
            except fmd.FmdError as e:                    # tile would not fit LDS: the previous tiling stays
                assert e.status == -6
        obank = oracle.new_bank(oracle.config(D, fast, slow), nch)
        ok = True
        for _ in range(int(rng.integers(1, 4))):
            B = int(rng.integers(1, 9))
            iq = rng.integers(0, 256, (nch, B * block), dtype=np.uint8)
            try:
                got = bank.demodulate_batch(iq)
            except fmd.FmdError as e:
                assert e.status in (-5, -6), (D, fast, slow, block, B, e)
                ok = False
                break
            for c in range(nch):
                exp = np.concatenate([oracle.demodulate(obank[c], iq[c, b * block:(b + 1) * block]) for b in range(B)])
                assert got[c].size == exp.size and np.array_equal(got[c], exp), (D, fast, slow, block, B, c)
            for c in range(nch):
                assert gpu_state(bank, c) == oracle.state_of(obank[c]), (D, fast, slow, block, B, c)
        bank.close()
        done += ok


def test_fuzz_streaming_kernel(fmd, oracle):
    """Downsample 2 and 4 with >= 8 channels run the register-streaming kernel (fmd_demod_stream_kernel: global memory ->
    registers, no LDS staging, tiles of 400 ... 1000 audio samples): random rates, 8 ... 40 channels, calls from a fraction of a
    tile to several tiles (and one reference-sized buffer), random / full-scale / near-silent / synthetic data, boxcar
    phases 0 and 2, the clamped spans at both ends of a call, state after every call."""
    n_cases = max(6, int(os.environ.get("FMD_FUZZ_CASES", "40")) // 3)
    rng = np.random.default_rng(int(os.environ.get("FMD_FUZZ_SEED", "20260101")) + 11)
    if not os.environ.get("FMD_LIB"):
        # the kernel under test is the one that runs: a bank of >= 8 channels reports the streaming kernel's tiling (larger
        # tiles -- 8 one-shot rounds per wave at downsample 4 since round 6 --, LDS for the discriminator samples only), a small
        # bank the LDS-DMA kernel's
        small, large = fmd.DemodBank(mkcfg(fmd, 4, 256000, 48000), 2).tiling(), fmd.DemodBank(mkcfg(fmd, 4, 256000, 48000), 8).tiling()
        assert large["audio_per_tile"] > 1.5 * small["audio_per_tile"] and large["lds_bytes"] < small["lds_bytes"] / 2, (small, large)
    for case in range(n_cases):
        D = int(rng.choice([2, 4]))
        slow = int(rng.choice(RATES))
        fast = max(slow, int(slow * rng.uniform(1.0, 9.0)) if rng.random() < 0.7 else int(rng.choice([256000, 500000, 300000, 64000, 250000])))
        nch = int(rng.integers(8, 41))
        blocks = []
        for k in range(int(rng.integers(2, 5))):
            big = rng.random() < 0.35
            n = 8 * int(rng.integers(4000, 40000)) if big else 8 * int(rng.integers(max(1, D // 2), 900))
            if case == 0 and k == 0:
                n = fmd.DEFAULT_BUF_LENGTH
            mode = rng.random()
            if mode < 0.5:
                blk = rng.integers(0, 256, (nch, n), dtype=np.uint8)
            elif mode < 0.7:
                blk = np.where(rng.integers(0, 2, (nch, n)) > 0, 255, 0).astype(np.uint8)
            elif mode < 0.85:
                blk = rng.integers(126, 131, (nch, n)).astype(np.uint8)
            else:
                blk = fmd.synth.synth_iq(nch, n, seed=int(rng.integers(1, 1 << 30)), amplitude=int(rng.integers(1, 121)))
            blocks.append(blk)
        try:
            check_stream(fmd, oracle, D, fast, slow, blocks, n_channels=nch)
        except fmd.FmdError as e:
            assert e.status in (-3, -5, -6), (D, fast, slow, nch, e)
        except AssertionError as e:
            raise AssertionError("case D=%d fast=%d slow=%d nch=%d sizes=%s: %s" % (D, fast, slow, nch, [b.shape[1] for b in blocks], e))
