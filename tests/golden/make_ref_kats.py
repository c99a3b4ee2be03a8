#!/usr/bin/env python3
"""Extract the known-answer vectors of the reference's three hot-path tests into JSON fixtures.

Runs only where /root/reference exists (the build container).  It copies DATA (the integer
vectors of `test_lowpass`, `test_demod`, `test_lowpass_real`, examples/simple_fm.rs:466-555,
provenance: osmocom `rtl_fm -f 92.5M -M fm -s 170k -A fast -r 32k -l 0`), never source text.
"""
import json, os, re, sys

REF = "/root/reference/examples/simple_fm.rs"
HERE = os.path.dirname(os.path.abspath(__file__))


def vec_after(lines, start, name):
    """Parse `let <name> = vec![ ... ];` that starts at or after 1-based line `start`."""
    i = start - 1
    while not re.search(r"let\s+%s\s*=\s*vec!\[" % re.escape(name), lines[i]):
        i += 1
    first = i + 1
    text = ""
    while True:
        text += lines[i]
        if "];" in lines[i]:
            break
        i += 1
    body = text[text.index("vec![") + 5: text.rindex("];")]
    vals = [int(t) for t in re.findall(r"-?\d+", body)]
    return vals, (first, i + 1)


def main():
    lines = open(REF).read().split("\n")
    lowpass, lp_lines = vec_after(lines, 466, "lowpass")
    buf_signed, bs_lines = vec_after(lines, 466, "buf_signed")
    lowpass2, lp2_lines = vec_after(lines, 514, "lowpass")
    demod_expected, de_lines = vec_after(lines, 514, "demod_expected")
    demodulated, dm_lines = vec_after(lines, 541, "demodulated")
    result, rs_lines = vec_after(lines, 541, "result")
    assert lowpass == lowpass2 and demod_expected == demodulated
    cfg = {"frequency": 94_900_000, "sample_rate": 170_000, "rate_resample": 32_000,
           "cite": "examples/simple_fm.rs:25-27"}
    out = {
        "ref_kat_lowpass.json": {
            "test": "test_lowpass", "cite": "examples/simple_fm.rs:466-511", "config": cfg,
            "input_buf_signed_i16": buf_signed, "input_lines": bs_lines,
            "expected_interleaved_i32": lowpass, "expected_lines": lp_lines},
        "ref_kat_demod.json": {
            "test": "test_demod", "cite": "examples/simple_fm.rs:514-538", "config": cfg,
            "input_interleaved_i32": lowpass2, "input_lines": lp2_lines,
            "expected_i16": demod_expected, "expected_lines": de_lines},
        "ref_kat_lowpass_real.json": {
            "test": "test_lowpass_real", "cite": "examples/simple_fm.rs:541-555", "config": cfg,
            "input_i16": demodulated, "input_lines": dm_lines,
            "expected_i16": result, "expected_lines": rs_lines},
    }
    for name, obj in out.items():
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(obj, f, indent=1)
        print(name, {k: (len(v) if isinstance(v, list) else v) for k, v in obj.items()})


if __name__ == "__main__":
    sys.exit(main())
