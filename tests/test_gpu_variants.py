"""GPU parity of the non-default kernel variants (same results by construction): the plain and index-arithmetic block
mappings (FMD_XCD), the general and the closed-form prologue (FMD_FAST=0 / 1; the default prefers the per-tile table), the
LDS-DMA kernel where the register-streaming one would run (FMD_STREAM=0), the generic fallback kernel
(FMD_FORCE_GENERIC).  These knobs exist in the
-DFMD_EXPERIMENT build only -- the shipped library reads no environment variable -- so every case re-runs itself in a
child process on libfmd_hip_exp.so (conftest.run_in_exp_child).  What the shipped library selects by itself (general
prologue for several phase classes, generic kernel beyond 16 of them) is covered in tests/test_gpu_parity.py."""
import numpy as np
import pytest

from conftest import run_in_exp_child
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


@pytest.mark.parametrize("env", [{"FMD_XCD": "0"}, {"FMD_XCD": "1"}, {"FMD_FAST": "0"}, {"FMD_FAST": "1"}, {"FMD_STREAM": "0"},
                                 {"FMD_FORCE_GENERIC": "1"}])
def test_kernel_variants_bit_exact(fmd, oracle, request, env):
    if run_in_exp_child(request, env):
        return
    cfgs = [CFG_24, CFG_REF, (4, 300000, 50000), (16, 62500, 31250)]
    if "FMD_STREAM" in env:                                   # downsample 2 and 4 with >= 8 channels: where the streaming kernel is the default
        for cfg in [(2, 500000, 32000), (4, 300000, 50000)]:
            check_stream(fmd, oracle, *cfg, blocks_for(fmd, 9, 3, seed=cfg[0]), n_channels=9)
        return
    for cfg in cfgs:
        check_stream(fmd, oracle, *cfg, blocks_for(fmd, 7, 3, seed=cfg[0]), n_channels=7)
        check_stream(fmd, oracle, *cfg, blocks_for(fmd, 3, 3, seed=cfg[0] + 1, n=8 * 517), n_channels=3)   # ragged small calls


def test_shipped_library_ignores_the_knobs(fmd, monkeypatch):
    """The release build must not let a stray variable swap the kernel or the tiling."""
    import os
    if os.environ.get("FMD_LIB"):
        pytest.skip("a tuning build is loaded")
    from test_gpu_parity import mkcfg
    ref = fmd.DemodBank(mkcfg(fmd, *CFG_24), 16).tiling()
    for k, v in {"FMD_XCD": "0", "FMD_KT": "7", "FMD_FORCE_GENERIC": "1", "FMD_F64_GUARD_LOG2": "-1"}.items():
        monkeypatch.setenv(k, v)
    assert fmd.DemodBank(mkcfg(fmd, *CFG_24), 16).tiling() == ref


def test_two_budget_tiling_plans_what_was_measured(fmd):
    """`choose_tiling` (csrc/fmd_api.cpp) plans twice: the most audio samples per tile that fit 20 KB of LDS (8 tiles per CU), and --
    for the rows whose vector work leaves the CU's SIMDs idle most of the time -- again among the tilings of 15.5 ... 17.3 KB
    (profiles/r05_experiments.md 9: -2 ... -4 % on those rows, the vector-heavier rows want the largest tile).  The figures in DESIGN.md
    and the committed profiles were measured with exactly these tilings; a planner change has to show up here."""
    import os
    if os.environ.get("FMD_LIB"):
        pytest.skip("a tuning build is loaded")
    from test_gpu_parity import mkcfg
    want = {(10, 240000, 32000): 101, (12, 192000, 32000): 105, (20, 200000, 48000): 91, (16, 150000, 32000): 106,   # re-planned (memory side)
            (6, 170000, 32000): 256, (5, 250000, 44100): 256, (7, 166666, 32000): 242, (8, 250000, 44100): 179,    # the largest that fits
            (64, 37500, 8000): 26, (14, 224000, 32000): 72}
    for cfg, kt in want.items():
        t = fmd.DemodBank(mkcfg(fmd, *cfg), 4096).tiling()
        assert t["audio_per_tile"] == kt and t["lds_bytes"] <= 20480 and t["block_threads"] == 256, (cfg, t)
        if cfg[0] >= 10 and cfg != (14, 224000, 32000) and cfg != (64, 37500, 8000):
            assert 15500 < t["lds_bytes"] <= 17700, (cfg, t)
