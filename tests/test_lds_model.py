"""tools/lds_model.py -- the model of how gfx950 serves ds_read_b128 that round 6 used to take the bank conflicts out of the sparse
operand reads -- against the committed measurements of tools/ldsbench.py (profiles/r06_ldsbench.jsonl, one MI355X): every pattern the
model calls conflict-free measured within 5 % of the contiguous read or better, every 2-way pattern 1.5 - 1.8 x, every 4-way one above
2.8 x; and the patterns the shipped kernels use are the conflict-free ones.  CPU test: data and arithmetic only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_model_orders_every_measured_pattern():
    import ldsbench
    import lds_model
    meas = {}
    with open(os.path.join(ROOT, "profiles", "r06_ldsbench.jsonl")) as f:
        for l in f:
            d = json.loads(l)
            meas[d["pattern"]] = d["vs_contiguous"]
    n = 0
    for name, (width, fn) in ldsbench.patterns().items():
        if width != 16 or name not in meas:
            continue
        p = lds_model.predict(fn)
        m = meas[name]
        if p == 1.0:
            assert m <= 1.05, (name, p, m)
        elif p == 2.0:
            assert 1.5 <= m <= 1.8, (name, p, m)
        elif p == 4.0:
            assert m >= 2.8, (name, p, m)
        n += 1
    assert n >= 25


def test_shipped_read_patterns_are_conflict_free_in_the_model():
    import lds_model
    # fused kernel, sparse forms (fmd_firdemod.hip fd_reg_body<SP>): columns of PC = 4 NG - 2 outputs of 16 bytes, the lanes of the odd
    # K quarters read their second half first
    for ng in (4, 6, 8):
        pc = 16 * (4 * ng - 2)
        assert lds_model.predict(lambda l, j, q, h: pc * j + 32 * q + 16 * (h ^ (q & 1))) == 1.0
        assert lds_model.predict(lambda l, j, q, h: pc * j + 32 * q + 16 * h) == 2.0          # round 5's order
    # stand-alone FIR, sparse form (fmd_fir.hip DIGITS = 3): the permuted tile image (fir_img) + the half swap, every chunk kc
    img = lambda a: a ^ ((a >> 3) & 0x60)
    for kc in range(4):
        assert lds_model.predict(lambda l, j, q, h: img(128 * (j + kc) + 32 * q + 16 * (h ^ (q & 1)))) == 1.0
    assert lds_model.predict(lambda l, j, q, h: 128 * j + 32 * q + 16 * h) == 4.0              # round 5
