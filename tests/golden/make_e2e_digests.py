#!/usr/bin/env python3
"""Freeze SHA-256 digests of the ORACLE's end-to-end output on the deterministic synthetic captures (the reference
ships no end-to-end vector: capture.bin is absent and its tests stop at the three per-pass KATs).  These digests do
not pin the oracle to the reference -- the KATs and the independent Python restatement do that -- they pin the
oracle (and through the parity tests the GPU path) against silent change.  Run from the repo root:
    python tests/golden/make_e2e_digests.py > tests/golden/e2e_digests.json"""
import hashlib, json, os, sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
import oracle_lib

CASES = [  # name, (D, fast, slow), channels, blocks, bytes per block, synth seed
    ("config1_cfg_ref_1ch_8x262144", (6, 170000, 32000), 1, 8, 262144, 0x05D50001),
    ("config2_cfg_2p4_1ch_4x262144", (10, 240000, 32000), 1, 4, 262144, 0x05D50002),
    ("cfg_2p4_16ch_3x65536", (10, 240000, 32000), 16, 3, 65536, 0x05D50003),
    ("odd_D7_4ch_5x30008", (7, 166666, 32000), 4, 5, 30008, 0x05D50004),
]


def case_input(fmd_synth, nch, blocks, nbytes, seed):
    return [fmd_synth.synth_iq(nch, nbytes, sample_offset=b * (nbytes // 2), seed=seed, amplitude=110 if b % 2 else 100)
            for b in range(blocks)]


def run(o, synth):
    out = {}
    for name, (D, fast, slow), nch, blocks, nbytes, seed in CASES:
        bank = o.new_bank(o.config(D, fast, slow), nch)
        h = hashlib.sha256()
        total = 0
        for iq in case_input(synth, nch, blocks, nbytes, seed):
            audio, lens = o.demodulate_batch(bank, iq, threads=1)
            for c in range(nch):
                h.update(np.ascontiguousarray(audio[c, :lens[c]]).astype("<i2").tobytes())
                total += int(lens[c])
        st = o.state_of(bank[nch - 1])
        out[name] = {"rates": [D, fast, slow], "channels": nch, "blocks": blocks, "block_bytes": nbytes, "seed": seed,
                     "audio_samples": total, "sha256_s16le": h.hexdigest(), "last_channel_state": st}
    return out


if __name__ == "__main__":
    import rtl_sdr_rs_amd as fmd
    print(json.dumps(run(oracle_lib.load(), fmd.synth), indent=1))
