"""GPU parity of the non-default kernel variants (same results by construction, selected by environment at
Demod creation): 64/128-thread tile blocks (FMD_NT), the plain and index-arithmetic block mappings (FMD_XCD), the general and
the table-driven prologue (FMD_FAST=0: general prologue, 1: closed-form geometry; the default prefers the per-tile table),
the generic fallback kernel (FMD_FORCE_GENERIC).  The register-streaming and persistent kernels of round 1
(measured slower) were removed in round 2."""
import numpy as np
import pytest

from test_gpu_parity import CFG_24, CFG_REF, check_stream

pytestmark = pytest.mark.gpu


def blocks_for(fmd, nch, ncalls, seed, n=None):
    n = n or fmd.DEFAULT_BUF_LENGTH
    rng = np.random.default_rng(seed)
    out = []
    for i in range(ncalls):
        if i % 2 == 0:
            out.append(fmd.synth.synth_iq(nch, n, sample_offset=i * (n // 2), amplitude=110))
        else:
            out.append(rng.integers(0, 256, (nch, n), dtype=np.uint8))
    return out


@pytest.mark.parametrize("env", [{"FMD_NT": "128"}, {"FMD_NT": "64"}, {"FMD_XCD": "0"}, {"FMD_XCD": "1"}, {"FMD_FAST": "0"}, {"FMD_FAST": "1"},
                                 {"FMD_FORCE_GENERIC": "1"}])
@pytest.mark.parametrize("cfg", [CFG_24, CFG_REF, (4, 300000, 50000), (16, 62500, 31250)])
def test_kernel_variants_bit_exact(fmd, oracle, monkeypatch, env, cfg):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    check_stream(fmd, oracle, *cfg, blocks_for(fmd, 7, 3, seed=cfg[0]), n_channels=7)
    check_stream(fmd, oracle, *cfg, blocks_for(fmd, 3, 3, seed=cfg[0] + 1, n=8 * 517), n_channels=3)   # ragged small calls
