#!/usr/bin/env python
"""Randomised differential run (development tool): the tests/test_fuzz.py loops with many iterations,
on the GPU library (default) or the host stand-in (--emu)."""
import _envopts  # noqa: F401  (SA_* environment -> library options, scripts/_envopts.py)
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _Env:
    def setenv(self, k, v):
        os.environ[k] = v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--emu", action="store_true")
    ap.add_argument("--seeds", type=int, default=20)
    args = ap.parse_args()
    if args.emu:
        from tests.emu import emu_api
        api = emu_api()
    else:
        from searcharray_amd import _lib
        api = _lib.api()
    from tests import test_fuzz, helpers
    from searcharray_amd import options as _options
    helpers._scope = _options.Scope()                     # (what tests/conftest.py's fixture gives a test: the scope set_opt writes into)
    fails = 0
    for seed in range(100, 100 + args.seeds):
        for fn in (test_fuzz.test_random_bm25_batches_pruned_and_exhaustive, test_fuzz.test_random_phrases_batches_and_slop, test_fuzz.test_random_slop_batches):
            try:
                fn(api, seed, _Env())
            except AssertionError as e:
                fails += 1
                print("FAIL", fn.__name__, seed, str(e)[:200], flush=True)
    print(f"fuzz done: {3 * args.seeds} runs, {fails} failures")


if __name__ == "__main__":
    main()
