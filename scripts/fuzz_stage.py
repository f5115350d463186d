#!/usr/bin/env python
"""The staged route's randomised differential test (tests/test_fuzz.py::test_random_bm25_batches_on_the_staged_route) over many seeds, on the
GPU library (default) or the host stand-in (--emu).  Development tool."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--emu", action="store_true")
    ap.add_argument("--first", type=int, default=1000)
    ap.add_argument("--seeds", type=int, default=100)
    ap.add_argument("--big", action="store_true", help="corpora of 50 K .. 400 K docs: thousands of stage tiles, co-walking groups on a full device")
    a = ap.parse_args()
    from tests import test_fuzz, helpers
    from searcharray_amd import options as _o
    helpers._scope = _o.Scope()
    if a.emu:
        from tests.emu import emu_api
        api = emu_api()
    else:
        from searcharray_amd import _lib
        api = _lib.api()
    if a.big:
        test_fuzz.STAGE_FUZZ_DOCS = (50_000, 400_000)
    bad = 0
    for seed in range(a.first, a.first + a.seeds):
        try:
            test_fuzz.test_random_bm25_batches_on_the_staged_route(api, seed)
        except AssertionError as e:
            bad += 1
            print("FAIL seed", seed, str(e)[:300], flush=True)
    print(f"{a.seeds} seeds from {a.first}: {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
