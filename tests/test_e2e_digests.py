"""End-to-end frozen digests (tests/golden/e2e_digests.json, made by tests/golden/make_e2e_digests.py from the oracle):
the CPU test pins the oracle and the closed-form model against silent change, the GPU test checks the HIP path
against the same frozen bytes -- nothing is recomputed by the thing under test's checker at run time."""
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_e2e_digests", os.path.join(HERE, "golden", "make_e2e_digests.py"))
mk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mk)
GOLD = json.load(open(os.path.join(HERE, "golden", "e2e_digests.json")))


def test_oracle_reproduces_frozen_digests(fmd, oracle):
    got = mk.run(oracle, fmd.synth)
    assert got == GOLD


@pytest.mark.gpu
@pytest.mark.parametrize("case", mk.CASES, ids=[c[0] for c in mk.CASES])
def test_gpu_reproduces_frozen_digests(fmd, case):
    name, (D, fast, slow), nch, blocks, nbytes, seed = case
    bank = fmd.DemodBank(fmd.DemodConfig(fast, fast, slow, D, max(1, (1 << 15) // (128 * D))), nch)
    h, total = hashlib.sha256(), 0
    for iq in mk.case_input(fmd.synth, nch, blocks, nbytes, seed):
        for a in bank.demodulate_batch(iq):
            h.update(np.ascontiguousarray(a).astype("<i2").tobytes())
            total += a.size
    g = GOLD[name]
    assert total == g["audio_samples"] and h.hexdigest() == g["sha256_s16le"]
    assert bank.get_state(nch - 1).as_dict() == g["last_channel_state"]
