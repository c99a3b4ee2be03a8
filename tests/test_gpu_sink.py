"""Pipelined multi-GPU sink (fmd_sink_*, new surface over the reference's receive -> mpsc -> process -> output
hand-off, examples/simple_fm.rs:55-60,114-127,150-156): >= 16 consecutive read_sync-sized buffers through a ring of
3 slots, channels split over two device parts (two GPUs where the box has them, else two banks on the one GPU),
must give exactly what the reference gives buffer by buffer, delivered in submission order."""
import numpy as np
import pytest

from test_gpu_parity import CFG_24, CFG_REF, mkcfg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,nch,depth,parts", [(CFG_REF, 11, 3, 2), (CFG_24, 64, 2, 1), (CFG_REF, 5, 4, 3)])
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
    s.close()
