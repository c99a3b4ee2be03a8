"""Client side of the rtl_tcp wire protocol as an IQ source for the GPU sink.

The reference ships the *server* (examples/rtl_tcp.rs): after accept it sends a 12-byte handshake -- b"RTL0",
tuner type (u32 big-endian), tuner gain count (u32 big-endian) (`send_handshake`, :691-697) -- then streams raw
interleaved u8 IQ exactly as `RtlSdr::read_sync` delivered it (`sender_loop`, :609-631), and accepts 5-byte
commands: one opcode byte + a big-endian u32/i32 parameter (`command_loop`, :633-689; opcodes 0x01-0x0e).
This module is the matching client, shaped like `read_sync` (src/lib.rs:153): fill the caller's buffer, return the
number of bytes written (a short count means the stream ended -- "samples lost" for the callers, simple_fm.rs:122).
Pure host code: sockets and bytes, no arithmetic.
"""
import socket
import struct

MAGIC = b"RTL0"

# opcodes of command_loop, examples/rtl_tcp.rs:659-675
CMD_SET_FREQUENCY = 0x01
CMD_SET_SAMPLE_RATE = 0x02
CMD_SET_GAIN_MODE = 0x03
CMD_SET_GAIN = 0x04
CMD_SET_FREQ_CORRECTION = 0x05
CMD_SET_IF_GAIN = 0x06
CMD_SET_TEST_MODE = 0x07
CMD_SET_AGC_MODE = 0x08
CMD_SET_DIRECT_SAMPLING = 0x09
CMD_SET_OFFSET_TUNING = 0x0A
CMD_SET_RTL_XTAL = 0x0B
CMD_SET_TUNER_XTAL = 0x0C
CMD_SET_GAIN_BY_INDEX = 0x0D
CMD_SET_BIAS_TEE = 0x0E


def pack_command(opcode, param):
    """5 bytes: opcode + big-endian 32-bit parameter (two's complement for negative values)."""
    return struct.pack(">BI", opcode & 0xFF, param & 0xFFFFFFFF)


def parse_handshake(data):
    if len(data) != 12 or data[:4] != MAGIC:
        raise ValueError("not an rtl_tcp handshake: %r" % (data[:12],))
    tuner_type, gain_count = struct.unpack(">II", data[4:])
    return tuner_type, gain_count


class RtlTcpSource:
    """`with RtlTcpSource(host, port) as src: n = src.read_sync(buf)`"""

    def __init__(self, host="127.0.0.1", port=1234, timeout=10.0):
        self.sock = socket.create_connection((host, port), timeout=timeout)
        self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        self.tuner_type, self.gain_count = parse_handshake(self._recv_exact(12))

    def _recv_exact(self, n):
        chunks, got = [], 0
        while got < n:
            b = self.sock.recv(n - got)
            if not b:
                break
            chunks.append(b)
            got += len(b)
        return b"".join(chunks)

    def read_sync(self, buf):
        """Fill `buf` (bytearray / writable memoryview / numpy uint8 array); returns bytes written."""
        view = memoryview(buf).cast("B")
        got = 0
        while got < len(view):
            n = self.sock.recv_into(view[got:], len(view) - got)
            if n == 0:
                break
            got += n
        return got

    def command(self, opcode, param):
        self.sock.sendall(pack_command(opcode, param))

    # the subset simple_fm's config_sdr (simple_fm.rs:217-229) uses, by name
    def set_center_freq(self, hz):
        self.command(CMD_SET_FREQUENCY, hz)

    def set_sample_rate(self, hz):
        self.command(CMD_SET_SAMPLE_RATE, hz)

    def set_tuner_gain_auto(self):
        self.command(CMD_SET_GAIN_MODE, 0)

    def set_bias_tee(self, on):
        self.command(CMD_SET_BIAS_TEE, 1 if on else 0)

    def close(self):
        try:
            self.sock.close()
        except OSError:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class RtlTcpSourceC:
    """The same source through the C ABI (fmd_rtltcp_*, include/fmd.h) -- what a non-Python host (the C++ mirror
    fm::RtlTcpSource, the Rust shim's RtlTcpSource) uses; same interface as RtlTcpSource above."""

    def __init__(self, host="127.0.0.1", port=1234, timeout=10.0):
        import ctypes as C
        from ._ffi import check, lib
        self._C, self._lib, self._check = C, lib(), check
        self._h = C.c_void_p()
        check(self._lib.fmd_rtltcp_open(host.encode(), port, int(timeout * 1000), C.byref(self._h)))
        t, g = C.c_uint32(), C.c_uint32()
        check(self._lib.fmd_rtltcp_info(self._h, C.byref(t), C.byref(g)))
        self.tuner_type, self.gain_count = t.value, g.value

    def read_sync(self, buf):
        view = memoryview(buf).cast("B")
        C = self._C
        n = C.c_size_t(0)
        addr = C.addressof((C.c_uint8 * len(view)).from_buffer(view))
        rc = self._lib.fmd_rtltcp_read_sync(self._h, addr, len(view), C.byref(n))
        self.partial = n.value                      # on FMD_ERR_IO (timeout / socket error): the bytes that did arrive
        self._check(rc)
        return n.value

    def command(self, opcode, param):
        self._check(self._lib.fmd_rtltcp_command(self._h, opcode & 0xFF, param & 0xFFFFFFFF))

    set_center_freq = RtlTcpSource.set_center_freq
    set_sample_rate = RtlTcpSource.set_sample_rate
    set_tuner_gain_auto = RtlTcpSource.set_tuner_gain_auto
    set_bias_tee = RtlTcpSource.set_bias_tee

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.fmd_rtltcp_close(self._h)
            self._h = self._C.c_void_p()

    def __del__(self):
        try:                                                 # (at interpreter shutdown the module globals may be gone already)
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def stream_fm(host, port, freq=94_900_000, rate=170_000, rate_resample=32_000, out=None, max_blocks=None):
    """simple_fm's receive()+process() (:89-170) with the dongle replaced by an rtl_tcp server and the Demod by
    the GPU one: tune with offset (optimal_settings :189-214), read DEFAULT_BUF_LENGTH blocks, write raw s16."""
    import sys
    import numpy as np
    from . import DEFAULT_BUF_LENGTH, Demod, optimal_settings
    out = out or sys.stdout.buffer
    radio, demod_cfg = optimal_settings(freq, rate, rate_resample)
    demod = Demod(demod_cfg)
    buf = np.empty(DEFAULT_BUF_LENGTH, dtype=np.uint8)
    with RtlTcpSource(host, port) as src:
        src.set_tuner_gain_auto()
        src.set_bias_tee(False)
        src.set_center_freq(radio.capture_freq)
        src.set_sample_rate(radio.capture_rate)
        n_blocks = 0
        while max_blocks is None or n_blocks < max_blocks:
            if src.read_sync(buf) < buf.size:          # short read: samples lost, exit (simple_fm.rs:122-125)
                break
            out.write(demod.demodulate(buf).tobytes())
            out.flush()
            n_blocks += 1
    return n_blocks


if __name__ == "__main__":
    import sys
    hp = sys.argv[1] if len(sys.argv) > 1 else "127.0.0.1:1234"
    h, _, p = hp.partition(":")
    stream_fm(h, int(p or 1234))
