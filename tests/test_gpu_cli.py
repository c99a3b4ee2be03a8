"""GPU check of the simple_fm file-mode equivalent (rtl-sdr-rs_amd/simple_fm_gpu):
stdout bytes must equal the oracle's file mode over the complete DEFAULT_BUF_LENGTH blocks."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "rtl-sdr-rs_amd", "simple_fm_gpu")


def oracle_file_mode(oracle, cfg, data, block):
    d = oracle.new(cfg)
    out = np.empty(data.size // 2 + 64, dtype=np.int16)
    n = oracle.lib.fmo_file_mode(C.byref(d), data.ctypes.data_as(C.POINTER(C.c_uint8)), data.size, block,
                                 out.ctypes.data_as(C.POINTER(C.c_int16)), out.size)
    assert n >= 0
    return out[:n]


def test_cli_file_mode_matches_oracle(fmd, oracle, tmp_path):
    if not os.path.exists(CLI):
        pytest.fail("simple_fm_gpu not built (run __graft_entry__.build())")
    N = fmd.DEFAULT_BUF_LENGTH
    # the synthetic capture at the example's own rate: 1.02 Msps, 170 kHz channel, 75 kHz deviation, 1 kHz tone
    data = fmd.synth.synth_iq(1, 5 * N + 4096, amplitude=60, dev_q32=int(75000 / 1020000 * 2**32), mod_period=1020)[0]
    path = tmp_path / "capture.bin"
    data.tofile(path)
    p = subprocess.run([CLI, str(path)], capture_output=True, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    got = np.frombuffer(p.stdout, dtype=np.int16)
    _, cfg = oracle.optimal_settings(94_900_000, 170_000)       # the example's constants, simple_fm.rs:25-27
    exp = oracle_file_mode(oracle, cfg, data, N)
    assert got.size == exp.size and np.array_equal(got, exp)
    assert b"dropped 4096 trailing bytes" in p.stderr           # EOF policy: complete blocks only
    # several blocks per launch (-b): byte-identical stdout, incl. a launch that is only partly filled
    for b in ("2", "4", "64"):
        pb = subprocess.run([CLI, "-b", b, str(path)], capture_output=True, timeout=120)
        assert pb.returncode == 0 and pb.stdout == p.stdout, b
        assert b"dropped 4096 trailing bytes" in pb.stderr
    # and through stdin with other rates
    p2 = subprocess.run([CLI, "-s", "240000", "-r", "48000", "-"], input=data.tobytes(), capture_output=True, timeout=120)
    assert p2.returncode == 0, p2.stderr.decode()
    _, cfg2 = oracle.optimal_settings(94_900_000, 240_000, 48_000)
    assert np.array_equal(np.frombuffer(p2.stdout, dtype=np.int16), oracle_file_mode(oracle, cfg2, data, N))


def test_cli_bank_mode_one_channel_per_file(fmd, oracle, tmp_path):
    """Several input files = one channel per file in one bank; every output file equals the oracle's file mode of
    its input over the blocks the shortest file allows."""
    N = fmd.DEFAULT_BUF_LENGTH
    blocks = [3, 2, 3]
    paths = []
    datas = []
    for k, nb in enumerate(blocks):
        d = fmd.synth.synth_iq(1, nb * N + 1000 * k, seed=100 + k, amplitude=40 + 30 * k)[0]
        p = tmp_path / ("cap%d.bin" % k)
        d.tofile(p)
        paths.append(str(p)); datas.append(d)
    prefix = str(tmp_path / "out")
    r = subprocess.run([CLI, "-o", prefix] + paths, capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    assert b"3 channels x 2 blocks" in r.stderr
    _, cfg = oracle.optimal_settings(94_900_000, 170_000)
    for k in range(3):
        got = np.fromfile("%s.%d.s16" % (prefix, k), dtype=np.int16)
        exp = oracle_file_mode(oracle, cfg, datas[k][:2 * N], N)
        assert got.size == exp.size and np.array_equal(got, exp), k
    # the same through the pipelined multi-GPU sink (-g N; device parts share the one GPU of a single-GPU box)
    for g in ("1", "2"):
        prefix2 = str(tmp_path / ("sink" + g))
        r2 = subprocess.run([CLI, "-g", g, "-o", prefix2] + paths, capture_output=True, timeout=120)
        assert r2.returncode == 0, r2.stderr.decode()
        assert ("3 channels x 2 blocks on %s device part(s)" % g).encode() in r2.stderr
        for k in range(3):
            a = np.fromfile("%s.%d.s16" % (prefix2, k), dtype=np.int16)
            b = np.fromfile("%s.%d.s16" % (prefix, k), dtype=np.int16)
            assert np.array_equal(a, b), (g, k)


def test_cli_live_mode_over_rtl_tcp(fmd, oracle):
    """-t host:port: the example's live path (receive + process, simple_fm.rs:89-170) with the dongle behind an rtl_tcp
    server: config_sdr's four settings arrive as commands, the audio equals the oracle's over the complete blocks, and
    the short read at the end of the stream ends the run with the example's message (:122-125)."""
    from test_rtl_tcp_source import FakeServer
    from rtl_sdr_rs_amd import rtl_tcp_source as rts
    N = fmd.DEFAULT_BUF_LENGTH
    data = fmd.synth.synth_iq(1, 3 * N + 5000, seed=77, amplitude=70, dev_q32=int(75000 / 1020000 * 2**32), mod_period=1020)[0]
    srv = FakeServer(data.tobytes())
    p = subprocess.run([CLI, "-t", "127.0.0.1:%d" % srv.port], capture_output=True, timeout=120)
    srv.thread.join(timeout=5)
    assert p.returncode == 0, p.stderr.decode()
    _, cfg = oracle.optimal_settings(94_900_000, 170_000)
    exp = oracle_file_mode(oracle, cfg, data[:3 * N], N)
    assert np.array_equal(np.frombuffer(p.stdout, dtype=np.int16), exp)
    assert b"Short read (5000 bytes), samples lost" in p.stderr and b"(3 loops)" in p.stderr
    for cmd in ((rts.CMD_SET_GAIN_MODE, 0), (rts.CMD_SET_BIAS_TEE, 0), (rts.CMD_SET_FREQUENCY, 95_155_000), (rts.CMD_SET_SAMPLE_RATE, 1_020_000)):
        assert cmd in srv.commands, cmd
