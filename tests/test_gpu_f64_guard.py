"""The one f64 sample of every reference call (Demod::polar_discriminant, simple_fm.rs:359,370-374) must not
depend on the last bits of the GPU's atan2: exact directions are decided with integers, every other sample within
the guard band of an integer is re-evaluated by the host libm and patched (include/fmd.h, fmd_demod_check).
Here the guard band is widened until EVERY generic sample takes that path (FMD_F64_GUARD_LOG2=-1: half-width 0.5)
and the kernel's own value is made wrong on purpose (FMD_F64_SKEW) -- both knobs of the -DFMD_EXPERIMENT build only; the
shipped library's band is fixed at 2^-20 (test_default_guard_band_is_quiet runs on it) -- so that
the results can only be right if the host patch works -- in the audio sample, in the carried partial sum, through
the host and the device entry points, and with several reference calls per launch."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "rtl-sdr-rs_amd", "libfmd_hip_exp.so")


def run_child(args, guard_log2=None, skew=None, extra_env=None):
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.update(extra_env or {})
    if guard_log2 is not None or skew is not None:
        # both knobs exist in the -DFMD_EXPERIMENT build only: the shipped library's guard band is 2^-20, fixed
        assert os.path.exists(EXP), "build() makes libfmd_hip_exp.so (make -C rtl-sdr-rs_amd/csrc exp)"
        env["FMD_LIB"] = EXP
    if guard_log2 is not None:
        env["FMD_F64_GUARD_LOG2"] = str(guard_log2)
    if skew is not None:
        env["FMD_F64_SKEW"] = str(skew)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "f64_child.py")] + [str(a) for a in args],
                       capture_output=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    return json.loads(p.stdout.decode().strip().splitlines()[-1])


def test_default_guard_band_is_quiet():
    r = run_child(["direct"])
    assert r["bad"] == 0 and r["state_bad"] == 0
    assert r["stats"]["guarded"] == 0 and r["stats"]["patched"] == 0     # 2^-20: nothing among 600 random samples


@pytest.mark.parametrize("kind", ["direct", "direct_device"])
def test_every_generic_sample_guarded_values_agree(kind):
    r = run_child([kind], guard_log2=-1)
    assert r["bad"] == 0 and r["state_bad"] == 0
    assert r["stats"]["guarded"] > 500 and r["stats"]["patched"] == 0    # ocml == libm after truncation: no patch needed


@pytest.mark.parametrize("kind", ["direct", "direct_device"])
def test_patch_repairs_a_wrong_gpu_value(kind):
    r = run_child([kind], guard_log2=-1, skew=3)
    assert r["bad"] == 0 and r["state_bad"] == 0
    assert r["stats"]["guarded"] > 500 and r["stats"]["patched"] == r["stats"]["guarded"]


@pytest.mark.parametrize("D,fast,slow,block_len", [(6, 170000, 32000, 0), (10, 240000, 32000, 0), (3, 48000, 44100, 0),
                                                   (6, 170000, 32000, 8 * 40), (10, 240000, 32000, 8 * 25),
                                                   (2, 500000, 8000, 0)])
def test_patch_inside_group_sums_and_carried_tail(D, fast, slow, block_len):
    """Groups of several discriminator samples: the patched value enters sum / R; (2, 500000, 8000): 62-sample
    groups, so the f64 sample of most calls lies in the carried partial sum (now_lpr) -> the state patch."""
    r = run_child(["stream", D, fast, slow, block_len], guard_log2=-1, skew=5)
    assert r["bad"] == 0 and r["state_bad"] == 0
    assert r["stats"]["guarded"] > 0 and r["stats"]["patched"] == r["stats"]["guarded"]


def test_fused_fir_kernel_guard_and_patch():
    """fmd_firdemod_*: the same guard on its one f64 sample per call, patched in the audio sample or the carried sum."""
    r = run_child(["firdemod"], guard_log2=-1)
    assert r["bad"] == 0 and r["state_bad"] == 0 and r["stats"]["guarded"] > 50 and r["stats"]["patched"] == 0
    r = run_child(["firdemod"], guard_log2=-1, skew=7)
    assert r["bad"] == 0 and r["state_bad"] == 0
    assert r["stats"]["guarded"] > 50 and r["stats"]["patched"] == r["stats"]["guarded"]


@pytest.mark.parametrize("D,fast,slow", [(6, 170000, 32000), (4, 48000, 48000)])
def test_sink_settles_guarded_samples_in_its_host_copy(D, fast, slow):
    """nothing exercised the sink's settle path (per-slot report buffer, host_out row offset of the device
    part, launch_seq of a slot that is no longer the newest launch).  With every f64 sample guarded AND wrong on the
    device, 9 buffers through 2 device parts and a ring of 3 are only right if each was patched in the slot's host copy.
    Run once without the skew too: then nothing needs a patch and the path must leave the audio alone."""
    r = run_child(["sink", D, fast, slow], guard_log2=-1)
    assert r["bad"] == 0 and r["stats"]["patched"] == 0 and 0 < r["stats"]["guarded"] <= r["expected_guarded"]
    r = run_child(["sink", D, fast, slow], guard_log2=-1, skew=9)
    assert r["bad"] == 0 and r["stats"]["patched"] == r["stats"]["guarded"] > 0      # measured by the sink, not assumed


@pytest.mark.parametrize("back", [1, 2])
@pytest.mark.parametrize("D,fast,slow,nch", [(6, 170000, 32000, 16), (10, 240000, 32000, 9), (4, 256000, 48000, 16), (5, 250000, 44100, 8),
                                             (2, 500000, 8000, 16), (6, 170000, 32000, 3)])
def test_check_behind_patches_the_older_launches_in_flight(D, fast, slow, nch, back):
    """fmd_demod_check_behind (round 6): enqueue launch n, settle launch n - back while the newer ones run.  With every f64 sample
    guarded and wrong on the device each launch's audio is only right if the patch went into ITS buffer although one or two newer
    launches had been enqueued; (2, 500000, 8000): 62-sample groups -- the sample mostly lies in the carried partial sum, which the next
    launch has consumed: the state it read is corrected (the ring keeps it while two newer launches are in flight) and the launches
    behind it run again, oldest first.  3 channels: the general prologue (no table), which posts too."""
    r = run_child(["pipelined", D, fast, slow, nch, back], guard_log2=-1)
    assert r["bad"] == 0 and r["state_bad"] == 0 and r["stats"]["patched"] == 0 and r["stats"]["guarded"] > 0
    r = run_child(["pipelined", D, fast, slow, nch, back], guard_log2=-1, skew=11)
    assert r["bad"] == 0 and r["state_bad"] == 0, r
    assert r["stats"]["guarded"] > 0 and r["stats"]["patched"] >= r["stats"]["guarded"] // 3, r    # (a replayed launch reports its samples again)


@pytest.mark.parametrize("back", [1, 2])
def test_check_behind_on_the_shipped_library_is_quiet_and_exact(back):
    r = run_child(["pipelined", 6, 170000, 32000, 64, back])
    assert r["bad"] == 0 and r["state_bad"] == 0 and r["stats"]["patched"] == 0


@pytest.mark.parametrize("back", [0, 1, 2])
def test_completion_points_under_event_ordering(back):
    """fmd_demod_set_event_ordering: every wait of the completion points goes to the event behind the launch -- fmd_demod_check reads
    the launches' report flags behind it (round 6: no head copy), fmd_demod_check_behind spins on the mailbox and falls back to the event.
    Every f64 sample guarded: once agreeing with the host (the light settle path), once wrong on the device (patches in the older launches'
    buffers, the carried sum through the replay chain); back = 0 is fmd_demod_check after every launch."""
    ev = {"FMD_TEST_EVENT_ORDERING": "1"}
    r = run_child(["pipelined", 6, 170000, 32000, 16, back], guard_log2=-1, extra_env=ev)
    assert r["bad"] == 0 and r["state_bad"] == 0 and r["stats"]["patched"] == 0 and r["stats"]["guarded"] > 0
    r = run_child(["pipelined", 2, 500000, 8000, 16, back], guard_log2=-1, skew=13, extra_env=ev)
    assert r["bad"] == 0 and r["state_bad"] == 0 and r["stats"]["guarded"] > 0 and r["stats"]["patched"] >= r["stats"]["guarded"] // 3, r

