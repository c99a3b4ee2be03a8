"""bench.py's launch paths.  `python bench.py --gpus N` must start N rank processes itself (`--gpus` was
parsed and ignored, so a plain `--gpus 8` measured one GPU).  CPU: the launcher really spawns N ranks and the whole
job fails loudly when there is no GPU (no silent CPU path, no silent single-rank run); a WORLD_SIZE / --gpus mismatch
is an error.  GPU: two ranks over gloo sharing the one GPU of the test box print ONE line with n_gpus == 2."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    return env


def test_gpus_flag_spawns_ranks_and_fails_loudly_without_gpu(fmd):
    if fmd.device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu test below")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--steps", "2", "--no-cpu"],
                       capture_output=True, env=clean_env(), timeout=600)
    err = p.stderr.decode()
    assert p.returncode != 0
    assert "rank exit codes" in err and err.count("no gfx950 device") >= 2, err[-2000:]   # both ranks ran and refused
    assert p.stdout.decode().strip() == ""


def test_a_dead_rank_takes_the_job_down(monkeypatch):
    """One rank exits with an error while another would sit in the rendezvous: the launcher ends the job (after a grace
    period for the others' own messages) instead of waiting for a collective timeout."""
    import importlib.util
    import time
    import types
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    real = subprocess.Popen

    def fake(cmd, env=None, stdout=None):
        code = "import time; time.sleep(120)" if env["RANK"] == "0" else "import sys; sys.exit(3)"
        return real([sys.executable, "-c", code], stdout=stdout)

    monkeypatch.setattr(bench.subprocess, "Popen", fake)
    t0 = time.time()
    rc = bench.spawn_ranks(types.SimpleNamespace(gpus=2))
    assert rc == 1 and time.time() - t0 < 60


def test_world_size_must_match_gpus():
    env = dict(clean_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2"], capture_output=True, env=env, timeout=300)
    assert p.returncode != 0 and b"WORLD_SIZE=2" in p.stderr


def run_bench(args):
    p = subprocess.run([sys.executable, BENCH] + args, capture_output=True, env=clean_env(), timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_on_one_box():
    common = ["--steps", "5", "--warmup", "2", "--settle", "10", "--no-cpu", "--no-extra", "--channels", "1024"]
    one = run_bench(common)
    two = run_bench(["--gpus", "2", "--backend", "gloo"] + common)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert len(two["per_gpu_msamples_per_s"]) == 2 and all(v > 0 for v in two["per_gpu_msamples_per_s"])
    assert two["config"]["parallelism"].startswith("channels sharded x2")
    assert "cpu_baseline" not in two and two["scaling"] == "weak"
    # both ranks share ONE GPU here, so the aggregate is about the one-rank rate (never 2x, never a silent 1-rank run)
    assert 0.4 * one["value"] < two["value"] < 1.6 * one["value"], (one["value"], two["value"])
    assert one["timing"]["regions"] >= 2 and one["timing"]["timed_ms_total"] >= 40.0
    for r in (one, two):
        assert 0.05 < r["roofline"]["frac"] < 1.0 and r["roofline"]["bound"] == "hbm"
