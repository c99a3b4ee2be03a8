"""rtl_tcp client-side IQ source against an in-process fake server that speaks the wire
format of the reference's examples/rtl_tcp.rs: 12-byte "RTL0" handshake (:691-697), raw u8 IQ (:609-631),
5-byte big-endian commands (:633-689)."""
import socket
import struct
import threading

import numpy as np
import pytest

from rtl_sdr_rs_amd import rtl_tcp_source as rts


class FakeServer:
    def __init__(self, payload, tuner_type=5, gain_count=29):
        self.payload, self.commands = payload, []
        self.lsock = socket.socket()
        self.lsock.bind(("127.0.0.1", 0))
        self.lsock.listen(1)
        self.port = self.lsock.getsockname()[1]
        self.hs = b"RTL0" + struct.pack(">II", tuner_type, gain_count)
        self.thread = threading.Thread(target=self.run, daemon=True)
        self.thread.start()

    def run(self):
        conn, _ = self.lsock.accept()
        conn.sendall(self.hs)
        conn.settimeout(0.2)
        sent = 0
        while sent < len(self.payload):
            conn.sendall(self.payload[sent:sent + 50000])
            sent += 50000
            try:
                while True:
                    c = conn.recv(5, socket.MSG_DONTWAIT)
                    if len(c) == 5:
                        self.commands.append(struct.unpack(">BI", c))
                    else:
                        break
            except (BlockingIOError, socket.timeout, OSError):
                pass
        try:
            while True:
                c = conn.recv(5)
                if len(c) < 5:
                    break
                self.commands.append(struct.unpack(">BI", c))
        except (socket.timeout, OSError):
            pass
        conn.close()
        self.lsock.close()


def test_handshake_commands_and_read_sync():
    rng = np.random.default_rng(1)
    payload = rng.integers(0, 256, 300000, dtype=np.uint8).tobytes()
    srv = FakeServer(payload)
    with rts.RtlTcpSource("127.0.0.1", srv.port) as src:
        assert (src.tuner_type, src.gain_count) == (5, 29)
        src.set_center_freq(95_155_000)
        src.set_sample_rate(1_020_000)
        src.command(rts.CMD_SET_FREQ_CORRECTION, -3)
        buf = np.empty(262144, dtype=np.uint8)
        assert src.read_sync(buf) == 262144                       # a full DEFAULT_BUF_LENGTH block
        assert buf.tobytes() == payload[:262144]
        assert src.read_sync(buf) == 300000 - 262144              # short read at end of stream
    srv.thread.join(timeout=5)
    assert (rts.CMD_SET_FREQUENCY, 95_155_000) in srv.commands
    assert (rts.CMD_SET_SAMPLE_RATE, 1_020_000) in srv.commands
    assert (rts.CMD_SET_FREQ_CORRECTION, 0xFFFFFFFD) in srv.commands   # i32 -3, big-endian two's complement


def test_c_abi_source_handshake_commands_and_read_sync(fmd):
    """The same through fmd_rtltcp_* (include/fmd.h): what the C++ mirror and the Rust shim's IqSource use.  Host code
    only -- it runs without a GPU."""
    rng = np.random.default_rng(2)
    payload = rng.integers(0, 256, 300000, dtype=np.uint8).tobytes()
    srv = FakeServer(payload, tuner_type=6, gain_count=28)
    with rts.RtlTcpSourceC("127.0.0.1", srv.port) as src:
        assert (src.tuner_type, src.gain_count) == (6, 28)
        src.set_center_freq(95_155_000)
        src.set_sample_rate(1_020_000)
        src.command(rts.CMD_SET_FREQ_CORRECTION, -3)
        buf = np.empty(262144, dtype=np.uint8)
        assert src.read_sync(buf) == 262144
        assert buf.tobytes() == payload[:262144]
        assert src.read_sync(buf) == 300000 - 262144              # short read at end of stream: not an error
        assert src.read_sync(buf) == 0
    srv.thread.join(timeout=5)
    assert (rts.CMD_SET_FREQUENCY, 95_155_000) in srv.commands
    assert (rts.CMD_SET_SAMPLE_RATE, 1_020_000) in srv.commands
    assert (rts.CMD_SET_FREQ_CORRECTION, 0xFFFFFFFD) in srv.commands


def test_c_abi_source_errors(fmd):
    import socket as so
    with so.socket() as s:                                       # a port nobody listens on
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with pytest.raises(fmd.FmdError) as ei:
        rts.RtlTcpSourceC("127.0.0.1", port, timeout=2.0)
    assert ei.value.status == -11                                 # FMD_ERR_IO
    srv = FakeServer(b"", tuner_type=0)
    srv.hs = b"HTTP/1.1 200"                                      # 12 bytes that are not a handshake
    with pytest.raises(fmd.FmdError) as ei:
        rts.RtlTcpSourceC("127.0.0.1", srv.port, timeout=2.0)
    assert ei.value.status == -11 and "RTL0" in str(ei.value)


def test_c_abi_source_timeout_reports_the_partial_count(fmd):
    """A server that stalls mid-block: read_sync fails with FMD_ERR_IO after the timeout and says how many bytes did
    arrive, so a caller that retries keeps the I/Q byte alignment of the stream."""
    stall = threading.Event()

    class Stalling(FakeServer):
        def run(self):
            conn, _ = self.lsock.accept()
            conn.sendall(self.hs + self.payload)
            stall.wait(10)
            conn.close()
            self.lsock.close()

    payload = bytes(range(256)) * 4 + b"xyz"                      # 1027 bytes, an odd count
    srv = Stalling(payload)
    with rts.RtlTcpSourceC("127.0.0.1", srv.port, timeout=0.4) as src:
        buf = np.zeros(4096, dtype=np.uint8)
        with pytest.raises(fmd.FmdError) as ei:
            src.read_sync(buf)
        assert ei.value.status == -11 and "timed out" in str(ei.value).lower()
        assert src.partial == len(payload) and buf[:len(payload)].tobytes() == payload
    stall.set()
    srv.thread.join(timeout=5)


def test_bad_handshake_rejected():
    with pytest.raises(ValueError):
        rts.parse_handshake(b"RTL1" + bytes(8))
    assert rts.parse_handshake(b"RTL0" + struct.pack(">II", 6, 29)) == (6, 29)
    assert rts.pack_command(0x04, -15) == bytes([4, 0xFF, 0xFF, 0xFF, 0xF1])


@pytest.mark.gpu
def test_stream_fm_over_rtl_tcp_matches_oracle(fmd, oracle):
    """End to end: rtl_tcp server -> client source -> GPU Demod == oracle on the same bytes."""
    import io
    N = fmd.DEFAULT_BUF_LENGTH
    data = fmd.synth.synth_iq(1, 3 * N + 1000, amplitude=60, dev_q32=int(75000 / 1020000 * 2**32), mod_period=1020)[0]
    srv = FakeServer(data.tobytes())
    sink = io.BytesIO()
    blocks = rts.stream_fm("127.0.0.1", srv.port, out=sink)
    srv.thread.join(timeout=5)
    assert blocks == 3
    _, cfg = oracle.optimal_settings(94_900_000, 170_000)
    exp, _ = oracle.demodulate_stream(cfg, data[:3 * N], N)
    assert np.array_equal(np.frombuffer(sink.getvalue(), dtype=np.int16), exp)
    assert (rts.CMD_SET_FREQUENCY, 95_155_000) in srv.commands     # offset tuning of optimal_settings (:195)


@pytest.mark.gpu
def test_three_rtl_tcp_streams_through_the_sink(fmd, oracle):
    """Three rtl_tcp servers (three dongles on other hosts) -> one pipelined sink with two device parts: the
    receive -> hand-off -> process -> output structure of simple_fm.rs:55-60 for several streams at once."""
    N = fmd.DEFAULT_BUF_LENGTH
    datas = [fmd.synth.synth_iq(1, 4 * N + 776 * k, seed=500 + k, amplitude=50 + 20 * k)[0] for k in range(3)]
    servers = [FakeServer(d.tobytes()) for d in datas]
    _, cfg = fmd.optimal_settings(94_900_000, 170_000)
    audio = [[] for _ in range(3)]

    def on_audio(seq, rows, status):
        assert status == 0
        for c in range(3):
            audio[c].append(rows[c])

    sink = fmd.Sink(cfg, 3, N, device_ids=[0, 1 % fmd.device_count()], depth=3, on_audio=on_audio)
    srcs = [rts.RtlTcpSource("127.0.0.1", servers[0].port)] + [rts.RtlTcpSourceC("127.0.0.1", s.port) for s in servers[1:]]   # both clients
    n = fmd.pump(srcs, sink)
    assert n == 4                                               # the 5th read is short on every stream
    for s in srcs:
        s.close()
    _, ocfg = oracle.optimal_settings(94_900_000, 170_000)
    for c in range(3):
        exp, _ = oracle.demodulate_stream(ocfg, datas[c][:4 * N], N)
        assert np.array_equal(np.concatenate(audio[c]), exp), c
    sink.close()
