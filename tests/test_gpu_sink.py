"""Pipelined multi-GPU sink (fmd_sink_*, new surface over the reference's receive -> mpsc -> process -> output
hand-off, examples/simple_fm.rs:55-60,114-127,150-156): >= 16 consecutive read_sync-sized buffers through a ring of
3 slots, channels split over two device parts (two GPUs where the box has them, else two banks on the one GPU),
must give exactly what the reference gives buffer by buffer, delivered in submission order."""
import numpy as np
import pytest

from test_gpu_parity import CFG_24, CFG_REF, mkcfg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,nch,depth,parts", [(CFG_REF, 11, 3, 2), (CFG_24, 64, 2, 1), (CFG_REF, 5, 4, 3),
                                                 (CFG_24, 67, 3, 8)])       # 8 device parts: the 8-GPU node's shape (uneven ranges)
def test_sink_matches_reference_buffer_by_buffer(fmd, oracle, cfg, nch, depth, parts):
    ndev = fmd.device_count()
    ids = [k % ndev for k in range(parts)]
    N = 65536
    got = []
    sink = fmd.Sink(mkcfg(fmd, *cfg), nch, N, device_ids=ids, depth=depth,
                    on_audio=lambda seq, rows, status: got.append((seq, rows, status)))
    assert sink.info()["n_devices"] == parts
    obank = oracle.new_bank(oracle.config(*cfg), nch)
    expected = []
    nbuf = 18
    for b in range(nbuf):
        iq = fmd.synth.synth_iq(nch, N, sample_offset=b * (N // 2), amplitude=100 + b)
        slot = sink.acquire()                               # page-locked; what read_sync would fill (src/lib.rs:153)
        slot[:] = iq
        sink.submit()
        assert sink.info()["in_flight"] <= depth
        exp, lens = oracle.demodulate_batch(obank, iq)
        expected.append([exp[c, :lens[c]].copy() for c in range(nch)])
    sink.drain()
    assert [g[0] for g in got] == list(range(nbuf))         # submission order
    for seq, rows, status in got:
        assert status == 0
        for c in range(nch):
            assert np.array_equal(rows[c], expected[seq][c]), (seq, c)
    sink.close()


def test_sink_rejects_misuse(fmd):
    cfg = mkcfg(fmd, *CFG_REF)
    with pytest.raises(fmd.FmdError) as ei:
        fmd.Sink(cfg, 4, 12)                                # nbytes % 8 (simple_fm.rs:286)
    assert ei.value.status == -2
    s = fmd.Sink(cfg, 4, 8192)
    with pytest.raises(fmd.FmdError):
        s.submit()                                          # nothing acquired
    s.acquire()
    with pytest.raises(fmd.FmdError):
        s.acquire()                                         # one slot at a time
    s.release()                                             # fmd_sink_release: un-acquire without submitting
    with pytest.raises(fmd.FmdError):
        s.release()                                         # nothing acquired any more
    s.acquire()                                             # ... and the sink is usable again
    s.release()
    s.close()


def test_pump_survives_a_short_read_and_consumer_errors_propagate(fmd, oracle):
    """pump() left the slot acquired after a short read (every later acquire failed); exceptions raised by
    on_audio inside the ctypes callback were printed and swallowed."""
    cfg = mkcfg(fmd, *CFG_REF)
    nch, N = 3, 8192

    class Src:
        def __init__(self, c, limit):
            self.c, self.n, self.limit = c, 0, limit

        def read_sync(self, buf):
            if self.n >= self.limit:
                buf[:100] = 0
                return 100                                  # short read: "samples lost" (simple_fm.rs:122)
            buf[:] = fmd.synth.synth_iq(1, N, sample_offset=self.n * (N // 2), first_channel=self.c)[0]
            self.n += 1
            return N

    got = []
    sink = fmd.Sink(cfg, nch, N, depth=2, on_audio=lambda seq, rows, status: got.append((seq, rows, status)))
    srcs = [Src(c, 4 if c != 1 else 3) for c in range(nch)]   # channel 1 runs dry first
    assert fmd.pump(srcs, sink) == 3 and [g[0] for g in got] == [0, 1, 2]
    obank = oracle.new_bank(oracle.config(*CFG_REF), nch)
    for b in range(3):
        iq = fmd.synth.synth_iq(nch, N, sample_offset=b * (N // 2))
        exp, lens = oracle.demodulate_batch(obank, iq)
        assert all(np.array_equal(got[b][1][c], exp[c, :lens[c]]) for c in range(nch))
    sink.acquire()                                          # still usable after the short read
    sink.release()

    def boom(seq, rows, status):
        raise ValueError("consumer failed on buffer %d" % seq)
    sink.on_audio = boom
    sink.push(fmd.synth.synth_iq(nch, N))
    with pytest.raises(ValueError, match="consumer failed"):
        sink.drain()
    sink.close()
