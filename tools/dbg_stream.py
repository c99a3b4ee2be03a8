import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, oracle_lib, rtl_sdr_rs_amd as fmd
o = oracle_lib.load()
D, fast, slow = 2, 1000000, 8000
rng = np.random.default_rng(D * 13 + 1)
nch = 9
cfg = fmd.DemodConfig(fast, fast, slow, D, 1)
bank = fmd.DemodBank(cfg, nch)
obank = o.new_bank(o.config(D, fast, slow), nch)
for i in range(4):
    n = int(rng.integers(2, 400)) * 8 + 16 * D
    blk = rng.integers(0, 256, (nch, n), dtype=np.uint8) if i % 2 else np.where(rng.integers(0, 2, (nch, n)) > 0, 255, 0).astype(np.uint8)
    got = bank.demodulate_batch(blk)
    exp, lens = o.demodulate_batch(obank, blk)
    print(i, n, "K", lens[0], "audio ok", all(np.array_equal(got[c], exp[c, :lens[c]]) for c in range(nch)), bank.tiling())
    for c in (0, 4, 8):
        a, b = bank.get_state(c).as_dict(), o.state_of(obank[c])
        if a != b: print("  ch", c, a, b)
