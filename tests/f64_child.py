"""Child process of tests/test_gpu_f64_guard.py: the guard band / skew knobs of the library are read from the
environment when a handle is created, and FMD_LIB selects the -DFMD_EXPERIMENT build, so each scenario runs in its
own interpreter.  Prints one JSON line.  TEST INFRASTRUCTURE."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

import oracle_lib
import rtl_sdr_rs_amd as fmd


def mkcfg(D, fast, slow):
    return fmd.DemodConfig(fast, fast, slow, D, max(1, (1 << 15) // (128 * D)))


def probe_states(n, seed):
    """(demod_pre, lp_now) pairs: the 8 exact directions at several magnitudes + random ones."""
    rng = np.random.default_rng(seed)
    pres, lps = [], []
    for v in [(1, 0), (0, 1), (-1, 0), (0, -1), (1, 1), (-1, 1), (1, -1), (-1, -1)]:
        for s in (1, 3, 100, 383):
            pres.append((1, 0)); lps.append((v[0] * s, v[1] * s))
            pres.append((v[0] * s, v[1] * s)); lps.append((v[0] * s, v[1] * s))
    while len(pres) < n:
        pres.append(tuple(int(x) for x in rng.integers(-512, 513, 2)))
        lps.append(tuple(int(x) for x in rng.integers(-384, 385, 2)))
    return pres[:n], lps[:n]


def scenario_direct(device_path):
    """fast == slow: every discriminator sample is an audio sample, so the f64 sample is out[0] itself; injected
    predecessors make it generic (non-special) for most channels."""
    o = oracle_lib.load()
    D, n = 4, 600
    cfg, ocfg = mkcfg(D, 48000, 48000), o.config(D, 48000, 48000)
    pres, lps = probe_states(n, 5)
    bank, obank = fmd.DemodBank(cfg, n), o.new_bank(ocfg, n)
    for c in range(n):
        bank.set_state(c, fmd.DemodState(prev_index=D - 1, now_lpr=0, prev_lpr_index=0, lp_now_re=lps[c][0],
                                         lp_now_im=lps[c][1], demod_pre_re=pres[c][0], demod_pre_im=pres[c][1]))
        obank[c].prev_index = D - 1
        obank[c].lp_now.re, obank[c].lp_now.im = lps[c]
        obank[c].demod_pre.re, obank[c].demod_pre.im = pres[c]
    rng = np.random.default_rng(6)
    bad = 0
    for call in range(3):
        iq = rng.integers(0, 256, (n, 64 + 8 * call), dtype=np.uint8)
        if call == 0:
            iq[:, 0:2] = 127
        exp, lens = o.demodulate_batch(obank, iq)
        if device_path:
            import torch
            d_iq = torch.from_numpy(iq).cuda()
            cap = bank.out_cap(iq.shape[1])
            d_out = torch.zeros((n, cap), dtype=torch.int16, device="cuda")
            bank.demodulate_device(d_iq.data_ptr(), iq.shape[1], d_out.data_ptr(), cap, None, None)
            bank.check()                                   # settles the guarded samples in d_out
            got = d_out.cpu().numpy()
            got = [got[c, :lens[c]] for c in range(n)]
        else:
            got = bank.demodulate_batch(iq)
        bad += sum(0 if np.array_equal(got[c], exp[c, :lens[c]]) else 1 for c in range(n))
    st_bad = sum(0 if bank.get_state(c).as_dict() == o.state_of(obank[c]) else 1 for c in range(0, n, 37))
    return {"bad": bad, "state_bad": st_bad, "stats": bank.f64_stats()}


def scenario_stream(D, fast, slow, block_len):
    """Ordinary streaming (groups of several samples, the f64 sample inside a group sum or in the carried tail),
    optionally with several reference calls per launch (set_block_len)."""
    o = oracle_lib.load()
    n, N = 24, 8 * 520
    cfg, ocfg = mkcfg(D, fast, slow), o.config(D, fast, slow)
    bank = fmd.DemodBank(cfg, n)
    ods = [o.new(ocfg) for _ in range(n)]
    rng = np.random.default_rng(D)
    bad = 0
    for call in range(6):
        nbytes = N if not block_len else block_len * int(rng.integers(1, 9))
        iq = rng.integers(0, 256, (n, nbytes), dtype=np.uint8)
        if block_len:
            bank.set_block_len(block_len)
        got = bank.demodulate_batch(iq)
        for c in range(n):
            if block_len:
                exp = np.concatenate([o.demodulate(ods[c], iq[c, b:b + block_len]) for b in range(0, nbytes, block_len)])
            else:
                exp = o.demodulate(ods[c], iq[c])
            bad += 0 if np.array_equal(got[c], exp) else 1
    st_bad = sum(0 if bank.get_state(c).as_dict() == o.state_of(ods[c]) else 1 for c in range(n))
    return {"bad": bad, "state_bad": st_bad, "stats": bank.f64_stats()}


def scenario_firdemod():
    """The fused FIR kernel's f64 sample (first filter output of every call): group sums of 52 / 5 samples, and a
    ratio at which the sample mostly lies in the carried tail."""
    o = oracle_lib.load()
    rng = np.random.default_rng(8)
    bad = st_bad = 0
    stats = {"guarded": 0, "patched": 0}
    for (T, M, fast, slow) in [(33, 4, 250000, 48000), (6, 6, 170000, 32000), (16, 8, 1000000, 8000)]:
        taps = np.ones(T, np.int16) if T == M else rng.integers(-2047, 2048, T).astype(np.int16)
        n = 7
        fd = fmd.FirDemodBank(taps, M, fast, slow, n)
        hs = [o.firdemod_new(taps, M, fd.shift, fast, slow) for _ in range(n)]
        for call in range(5):
            iq = rng.integers(0, 256, (n, 8 * int(rng.integers(60, 900))), dtype=np.uint8)
            got = fd.demodulate_batch(iq)
            for c in range(n):
                bad += 0 if np.array_equal(got[c], o.firdemod(hs[c], iq[c])) else 1
        for c in range(n):
            a, b = fd.get_state(c).as_dict(), o.firdemod_state(hs[c])
            st_bad += 0 if (a["now_lpr"], a["demod_pre"]) == (b["now_lpr"], b["demod_pre"]) else 1
        s = fd.f64_stats()
        stats["guarded"] += s["guarded"]; stats["patched"] += s["patched"]
    return {"bad": bad, "state_bad": st_bad, "stats": stats}


def scenario_sink(D, fast, slow):
    """The sink's settle path (fmd_sink.cpp complete_oldest -> fmd_internal_resolve_exc with the slot's own report
    buffer, a host_out row offset and the slot's launch_seq): 2 device parts, ring of 3 slots, every call's f64 sample
    guarded (and, with FMD_F64_SKEW, wrong on the device), so every delivered buffer is only right if it was patched in
    the slot's HOST copy -- for the right rows of the right part -- while later launches are already enqueued."""
    o = oracle_lib.load()
    n, N, nbuf = 11, 8 * 650, 9
    cfg, ocfg = mkcfg(D, fast, slow), o.config(D, fast, slow)
    got = []
    sink = fmd.Sink(cfg, n, N, device_ids=[0, 0], depth=3, on_audio=lambda seq, rows, status: got.append((seq, rows, status)))
    obank = o.new_bank(ocfg, n)
    rng = np.random.default_rng(D + 40)
    exps = []
    for b in range(nbuf):
        iq = rng.integers(0, 256, (n, N), dtype=np.uint8)
        sink.push(iq)
        e, l = o.demodulate_batch(obank, iq)
        exps.append([e[c, :l[c]].copy() for c in range(n)])
    sink.drain()
    bad = sum(0 if (st == 0 and np.array_equal(rows[c], exps[seq][c])) else 1 for seq, rows, st in got for c in range(n))
    bad += 0 if [g[0] for g in got] == list(range(nbuf)) else 1000
    stats = sink.f64_stats()                                   # the parts' real counters (fmd_sink_f64_stats)
    sink.close()
    return {"bad": bad, "state_bad": 0, "stats": stats, "expected_guarded": n * nbuf}


def scenario_pipelined(D, fast, slow, nch, back=1):
    """fmd_demod_check_prev: launch n is enqueued, THEN launch n - 1 is settled and read -- two launches in flight, the patch has to
    land in the OLDER one's output buffer (and, where the f64 sample of launch n - 1 lies in the sum it carried into launch n, in
    the state launch n read, with launch n run again).  Every launch has its own input and output buffer, as the contract asks;
    the last launch is settled by fmd_demod_check.  Channels >= 8 and 4096-byte calls: the table-form tile / streaming kernels."""
    import torch
    o = oracle_lib.load()
    N, ncalls = 8 * 512, 7
    cfg, ocfg = mkcfg(D, fast, slow), o.config(D, fast, slow)
    bank, obank = fmd.DemodBank(cfg, nch), o.new_bank(ocfg, nch)
    if os.environ.get("FMD_TEST_EVENT_ORDERING"):
        bank.set_event_ordering(True)                        # every wait goes to the launch's event (the report flags are read behind it)
    rng = np.random.default_rng(D + 90)
    cap = bank.out_cap(N)
    ins, outs, exps = [], [], []
    bad = 0
    kernels = set()

    def settle_and_compare(k):
        got = outs[k].cpu().numpy()
        e, l = exps[k]
        return sum(0 if np.array_equal(got[c, :l[c]], e[c, :l[c]]) else 1 for c in range(nch))

    for call in range(ncalls):
        iq = rng.integers(0, 256, (nch, N), dtype=np.uint8)
        exps.append(o.demodulate_batch(obank, iq))
        ins.append(torch.from_numpy(iq).cuda())
        outs.append(torch.zeros((nch, cap), dtype=torch.int16, device="cuda"))
        bank.demodulate_device(ins[-1].data_ptr(), N, outs[-1].data_ptr(), cap, None, None)
        kernels.add(bank.last_kernel())
        bank.check_behind(back)                            # launch `call - back` is final now; the newer ones may still run
        if call >= back:
            bad += settle_and_compare(call - back)
    bank.check()
    for k in range(max(0, ncalls - back), ncalls):
        bad += settle_and_compare(k)
    st_bad = sum(0 if bank.get_state(c).as_dict() == o.state_of(obank[c]) else 1 for c in range(nch))
    return {"bad": bad, "state_bad": st_bad, "stats": bank.f64_stats(), "kernels": sorted(kernels), "launches": ncalls * nch}


if __name__ == "__main__":
    kind = sys.argv[1]
    if kind == "direct":
        res = scenario_direct(False)
    elif kind == "direct_device":
        res = scenario_direct(True)
    elif kind == "firdemod":
        res = scenario_firdemod()
    elif kind == "sink":
        res = scenario_sink(*(int(x) for x in sys.argv[2:5]))
    elif kind == "pipelined":
        res = scenario_pipelined(*(int(x) for x in sys.argv[2:7]))
    else:
        D, fast, slow, bl = (int(x) for x in sys.argv[2:6])
        res = scenario_stream(D, fast, slow, bl)
    print(json.dumps(res))
