"""The reference's OWN known-answer vectors run straight through the HIP path (C ABI, include/fmd.h) -- no oracle in between.

examples/simple_fm.rs:466-555 holds three chained tests (vectors captured from osmocom rtl_fm, :461,468,516): `test_lowpass`
(512 centred i16 `buf_signed` :481-507 -> 42 complex :469-475), `test_demod` (42 complex -> 42 i16 :525-530, first sample through
the f64 path against a zero predecessor) and `test_lowpass_real` (42 i16 -> `[2588, 4030, -1212, -3430, 2585, 2110, -6110]` :549),
each on a fresh `Demod` at the shipped example's configuration (downsample 6, 170 kHz -> 32 kHz, :25-27,189-214).  The other GPU
parity tests compare the kernels with the oracle, which `tests/test_oracle_kat.py` pins to these vectors: sound but transitive.
Here the chain is closed directly: `buf_signed` (range -85 ... 86) is what rotate_90 (:276-299) + `as i16 - 127` (:258) left of the
raw read_sync bytes, and both steps are invertible -- plain byte = v + 127, negated byte (`255 - x`) = 128 - v, bytes 2/3 and 6/7
of every 8 swapped -- so the RAW buffer the reference's `demodulate` would have been handed exists, and
  (i)   `demodulate(raw)` on a fresh bank at cfg-ref must return the seven audio samples of :549,
  (ii)  with rate_resample == rate_out (resampler divisor 1: every discriminator sample is emitted) the 42 values of :525-530,
  (iii) the tapped FIR with six all-ones taps, decimate 6, the 42 complex values of :469-475.
The fixtures under tests/golden/ are the reference's integer vectors extracted as data (tests/golden/make_ref_kats.py); nothing
here reads /root/reference.
"""
import json
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def raw_bytes_of(buf_signed):
    """Invert `as i16 - 127` (:258) and the scalar rotate_90 (:286-295) on the centred samples of test_lowpass.

    rotate_90 maps raw b0..b7 to [b0, b1, 255-b3, b2, 255-b4, 255-b5, b7, 255-b6]; centring subtracts 127."""
    s = np.asarray(buf_signed, dtype=np.int64).reshape(-1, 8)
    raw = np.empty_like(s)
    raw[:, 0] = s[:, 0] + 127
    raw[:, 1] = s[:, 1] + 127
    raw[:, 3] = 128 - s[:, 2]          # 255 - b3 - 127 = s2
    raw[:, 2] = s[:, 3] + 127
    raw[:, 4] = 128 - s[:, 4]
    raw[:, 5] = 128 - s[:, 5]
    raw[:, 7] = s[:, 6] + 127
    raw[:, 6] = 128 - s[:, 7]
    assert raw.min() >= 0 and raw.max() <= 255, "buf_signed is the image of real u8 samples"
    return raw.reshape(-1).astype(np.uint8)


def forward_check(raw, buf_signed):
    """The inversion above, checked by running the reference's two steps forward in numpy (so that a wrong inverse cannot
    be compensated by a kernel bug): rotate (:286-295), centre (:258)."""
    b = raw.astype(np.int64).reshape(-1, 8)
    rot = np.stack([b[:, 0], b[:, 1], 255 - b[:, 3], b[:, 2], 255 - b[:, 4], 255 - b[:, 5], b[:, 7], 255 - b[:, 6]], axis=1)
    assert np.array_equal((rot - 127).reshape(-1), np.asarray(buf_signed, dtype=np.int64))


@pytest.fixture(scope="module")
def kat_raw():
    g = gold("ref_kat_lowpass.json")
    raw = raw_bytes_of(g["input_buf_signed_i16"])
    forward_check(raw, g["input_buf_signed_i16"])
    assert raw.size == 512
    return raw


def ref_demod_config(fmd, rate_resample=None):
    g = gold("ref_kat_demod.json")["config"]
    _, d = fmd.optimal_settings(g["frequency"], g["sample_rate"], g["rate_resample"] if rate_resample is None else rate_resample)
    assert (d.downsample, d.rate_out) == (6, 170000)         # simple_fm.rs:190,207-210
    return d


def test_end_to_end_seven_audio_samples(fmd, kat_raw):
    """(i) simple_fm.rs:549 -- one 512-byte `demodulate` on a fresh Demod at the reference's own configuration."""
    exp = np.array(gold("ref_kat_lowpass_real.json")["expected_i16"], dtype=np.int16)
    cfg = ref_demod_config(fmd)
    d = fmd.Demod(cfg)
    got = d.demodulate(kat_raw)
    assert got.tolist() == exp.tolist() == [2588, 4030, -1212, -3430, 2585, 2110, -6110]
    st = d.get_state(0).as_dict()
    # what the three reference tests leave behind: boxcar phase 256 % 6 (test_lowpass), the partial sum of its last four
    # samples, the last decimated sample as predecessor (test_demod), and the resampler's carry (test_lowpass_real)
    lp = np.array(gold("ref_kat_demod.json")["input_interleaved_i32"], dtype=np.int64).reshape(-1, 2)
    sig = np.array(gold("ref_kat_lowpass.json")["input_buf_signed_i16"], dtype=np.int64).reshape(-1, 2)
    assert st["prev_index"] == 4
    assert st["lp_now"] == sig[252:].sum(axis=0).tolist()
    assert st["demod_pre"] == lp[-1].tolist()
    assert (st["prev_lpr_index"], st["now_lpr"]) == (154000, 7139)
    d.close()
    # the same through the batched entry with the vector in every channel of a bank (one phase class, the table prologue)
    bank = fmd.DemodBank(cfg, 9)
    outs = bank.demodulate_batch(np.tile(kat_raw, (9, 1)))
    assert all(o.tolist() == exp.tolist() for o in outs)
    bank.close()


def test_discriminator_values(fmd, kat_raw):
    """(ii) simple_fm.rs:525-530 -- with rate_resample == rate_out `low_pass_real` divides by 170000 / 170000 = 1 and emits after
    every sample (:411-420), so the audio IS the discriminator output: d[0] = 0 by the f64 path against (0, 0), then 41 x fast_atan2."""
    exp = gold("ref_kat_demod.json")["expected_i16"]
    d = fmd.Demod(ref_demod_config(fmd, rate_resample=170000))
    got = d.demodulate(kat_raw)
    assert got.tolist() == exp and len(exp) == 42 and exp[0] == 0
    d.close()


def test_boxcar_values_through_the_tapped_fir(fmd, kat_raw):
    """(iii) simple_fm.rs:469-475 -- `low_pass_complex` at downsample 6 is the tapped FIR with six all-ones taps, decimate 6:
    42 complex sums of 256 samples (the last four stay in the filter's history, as prev_index = 4 says in the reference)."""
    exp = np.array(gold("ref_kat_lowpass.json")["expected_interleaved_i32"], dtype=np.int32).reshape(-1, 2)
    fir = fmd.FirBank(np.ones(6, dtype=np.int16), 6, 1)
    got = fir.filter_batch(kat_raw[None, :])
    assert got.shape == (1, 42, 2) and np.array_equal(got[0], exp)
    fir.close()
    # and the fused operator with those taps and no normalisation is the reference chain again: the seven samples of :549
    fd = fmd.FirDemodBank(np.ones(6, dtype=np.int16), 6, 170000, 32000, 1, shift=0)
    out = fd.demodulate_batch(kat_raw[None, :])
    assert out[0].tolist() == gold("ref_kat_lowpass_real.json")["expected_i16"]
    st = fd.get_state(0).as_dict()
    assert (st["prev_lpr_index"], st["now_lpr"], st["demod_pre"]) == (154000, 7139, exp[-1].tolist())
    fd.close()


def test_chained_calls_continue_the_vectors(fmd, kat_raw):
    """The KAT buffer cut into four calls (state carried, :232-239).  A cut moves the f64 sample (first sample of every call,
    :359), but at rate_resample == rate_out every OTHER discriminator value must still be the reference's: 42 - 3 of them."""
    exp = gold("ref_kat_demod.json")["expected_i16"]
    d = fmd.Demod(ref_demod_config(fmd, rate_resample=170000))
    cuts = [0, 96, 200, 360, 512]                              # multiples of 8 (rotate_90's period, :284)
    got, first = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        first.append(len(got))
        got += d.demodulate(kat_raw[a:b]).tolist()
    assert len(got) == 42
    same = [i for i in range(42) if i not in first[1:]]
    assert [got[i] for i in same] == [exp[i] for i in same]
    # the moved f64 samples are polar_discriminant (:370-374) of the reference's OWN decimated samples (:469-475), which plain
    # libm arithmetic restates in three lines: c = a * conj(b), (atan2(c.im, c.re) / PI * 16384) as i32
    lp = np.array(gold("ref_kat_demod.json")["input_interleaved_i32"], dtype=np.int64).reshape(-1, 2)
    for i in first[1:]:
        (ar, ai), (br, bi) = lp[i].tolist(), lp[i - 1].tolist()
        cr, ci = ar * br + ai * bi, ai * br - ar * bi
        assert got[i] == int(math.atan2(ci, cr) / math.pi * 16384.0), i
    d.close()
