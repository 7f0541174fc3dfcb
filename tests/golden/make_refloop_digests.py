#!/usr/bin/env python3
"""Generates tests/golden/refloop_digests.json.  Run in the build container (needs oracle/_ref/liboracle_refloop.so, which
oracle/Makefile builds only where /root/reference is mounted).

For every case of tests/refloop_cases.py the REFLOOP build of the oracle - the restatement with the real ULCi_GetWindowCtrl,
ULCi_CalculateNoiseLogSpectrum, ULCi_CalculatePsychoacoustics, ULCi_GetNoiseQ and ULCi_GetHFExtParams called at the
reference's own call sites - encodes the stream; recorded are the per-site digests of everything that went into and came
out of those calls (FNV-1a 64, oracle/orc_encoder.c site_*), the call counts, and the sha256 of the stream it wrote.
Data only: numbers produced by running the reference's compiled functions, no source text."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from refloop_cases import CASES, run_case, load_refloop  # noqa: E402


def main():
    lib = load_refloop()
    assert lib is not None and lib.orc_site_is_refloop() == 1, "needs oracle/_ref/liboracle_refloop.so (make -C oracle ref)"
    out = {}
    for tag in CASES:
        r = run_case(lib, tag)
        out[tag] = {"sites": {k: "%016x" % v for k, v in r["dig"].items()}, "calls": r["cnt"], "stream_sha256": r["sha"],
                    "blocks": r["blocks"], "bytes": r["bytes"]}
        print(tag, out[tag]["calls"], r["bytes"])
    json.dump(out, open(os.path.join(HERE, "refloop_digests.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
