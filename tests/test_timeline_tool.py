"""tools/timeline.py's analysis on a synthetic launch with a known answer (no GPU): 2 CUs x 8 slots, every slot runs 5 blocks of
1000 clocks (600 staging) with 100 empty clocks between them -- the tool has to find the turn-over gap, the phases and the residency."""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_analyse():
    src = open(os.path.join(ROOT, "tools", "timeline.py")).read()
    body = src.split("def analyse", 1)[1].split("\ndef main", 1)[0]       # (the module imports torch and the library at top level)
    ns = {"np": np}
    exec("def analyse" + body, ns)
    return ns["analyse"]


def test_analysis_recovers_gap_phases_and_residency():
    analyse = load_analyse()
    rows = []
    for cu in range(2):
        for slot in range(8):
            t = 1000 * cu + slot * 37
            for k in range(5):
                for w in range(4):
                    rows.append([t & 0xFFFFFFFF, 0, 600, 1000 + w, (cu << 8) | (w << 4) | slot, 0, cu * 100 + slot, k | (k << 16)])
                t += 1000 + 3 + 100
    out = analyse(np.array(rows, dtype=np.uint32), 1024)
    assert out["cus"] == 2 and out["simds"] == 8 and out["blocks"] == 80
    assert abs(out["slot_turnover_gap_clk"]["mean"] - 100.0) < 1e-9 and out["max_resident_blocks_last_cu"] == 8
    assert abs(out["staging_clk"]["mean"] - 600.0) < 1e-9 and abs(out["compute_clk"]["mean"] - 401.5) < 1e-9
    # 8 slots, each busy 5 * 1003 of its 5 * 1003 + 4 * 100 clocks, seen over the CU's span (the slots start 37 clocks apart)
    assert 7.0 < out["resident_blocks_per_cu"]["mean"] < 7.2
    assert 4.1 < out["staging_blocks_per_cu"]["mean"] < 4.3


def test_records_of_unwritten_waves_are_ignored():
    analyse = load_analyse()
    rec = np.zeros((64, 8), dtype=np.uint32)
    for w in range(4):
        rec[w] = [500, 0, 300, 900, (w << 4), 0, 7, 0]
    out = analyse(rec, 1024)
    assert out["tile_waves"] == 4 and out["blocks"] == 1
