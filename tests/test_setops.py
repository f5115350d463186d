"""Sorted-array primitives (the reference's snp_ops) on the device vs the CPU oracle, the
reference's scenario dicts (test/test_snp_ops.py) and its captured fixture arrays (tests/golden)."""
import numpy as np
import pytest

from oracle import refimpl as O
from searcharray_amd import ops
from tests.helpers import load_golden

u64 = lambda x: np.asarray(x, dtype=np.uint64)  # noqa: E731
HEADER = np.uint64(0xFFFFFFFFFFFC0000)


def _rand_sorted(rng, n, hi, dups):
    a = rng.integers(0, hi, n).astype(np.uint64)
    if not dups:
        a = np.unique(a)
    return np.sort(a)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("dups", [False, True])
def test_random_against_oracle(api, seed, dups):
    rng = np.random.default_rng(seed)
    lhs = _rand_sorted(rng, int(rng.integers(1, 3000)), 4000, dups)
    rhs = _rand_sorted(rng, int(rng.integers(1, 3000)), 4000, dups)
    for mask in (O.ALL_BITS, np.uint64(0xFFFFFFFFFFFFFFF8)):
        for drop in (True, False):
            a, b = ops.intersect(lhs, rhs, mask=mask, drop_duplicates=drop, api=api)
            c, d = O.intersect(lhs, rhs, mask=mask, drop_duplicates=drop)
            assert np.array_equal(a, c) and np.array_equal(b, d), (seed, dups, int(mask), drop)
        # With duplicate masked values on the rhs the reference's galloping pointer may sit on ANY
        # member of the equal run when a pair is recorded (path dependent); the device reports the
        # first member.  Same lhs indices, same matched values; identical indices when the masked
        # rhs values are unique -- which is how the reference calls these (roaringish headers).
        unique_rhs = len(np.unique(rhs & mask)) == len(rhs)
        a, b = ops.adjacent(lhs, rhs, mask=mask, api=api)
        c, d = O.adjacent(lhs, rhs, mask=mask)
        assert np.array_equal(a, c), ("adjacent", seed, dups)
        assert np.array_equal(rhs[b.astype(int)] & mask, rhs[d.astype(int)] & mask)
        if unique_rhs:
            assert np.array_equal(b, d)
        got = ops.intersect_with_adjacents(lhs, rhs, mask=mask, api=api)
        want = O.intersect_with_adjacents(lhs, rhs, mask=mask)
        for k_, (g_, w_) in enumerate(zip(got, want)):
            if k_ in (0, 2) or unique_rhs:
                assert np.array_equal(g_, w_), ("iwa", seed, dups, k_)
            else:
                assert np.array_equal(rhs[g_.astype(int)] & mask, rhs[w_.astype(int)] & mask), ("iwa", seed, dups, k_)
    for drop in (False, True):
        assert np.array_equal(ops.merge(lhs, rhs, drop_duplicates=drop, api=api), O.merge(lhs, rhs, drop_duplicates=drop))
    li, ri = np.unique(lhs), np.unique(rhs)
    lc, rc = rng.integers(0, 9, len(li)).astype(np.float32), rng.integers(0, 9, len(ri)).astype(np.float32)
    gi, gc = ops.sort_merge_counts(li, lc, ri, rc, api=api)
    wi, wc = O.sort_merge_counts(li, lc, ri, rc)
    assert np.array_equal(gi, wi) and np.array_equal(gc, wc)
    ids = np.sort(rng.integers(0, 300, 2500)).astype(np.uint64)
    pay = rng.integers(0, 2 ** 40, 2500).astype(np.uint64)
    for fn, ofn in ((ops.popcount_reduce_at, O.popcount_reduce_at), (ops.key_sum_over, O.key_sum_over)):
        gi, gc = fn(ids, pay % np.uint64(50), api=api)
        wi, wc = ofn(ids, pay % np.uint64(50))
        assert np.array_equal(gi, wi) and np.array_equal(gc, wc)
    words = (ids << np.uint64(36)) | ((pay % np.uint64(5)) << np.uint64(18)) | np.uint64(1)
    assert np.array_equal(ops.payload_slice(words, 0xFFFFC0000, 0, 1 << 18, api=api),
                          O.payload_slice(words, 0xFFFFC0000, 0, 1 << 18))


def test_reference_scenarios(api):
    """scenario dicts from reference test/test_snp_ops.py:96-154, :457-522, :537-548"""
    lhs, rhs = u64([1, 1, 2, 2, 3, 3, 4, 4, 5, 5]), u64([1, 2, 2, 10])
    li, ri = ops.intersect(lhs, rhs, api=api)
    assert (lhs[li.astype(int)] == [1, 2]).all()
    lhs = u64([0x1F, 0x2F, 0x3F, 0x4F, 0x5F, 0x6F, 0x7F, 0x8F, 0x9F, 0xAF])
    rhs = u64([0x2F, 0x4F, 0x6F, 0x8F, 0xAF])
    li, ri = ops.intersect(lhs, rhs, mask=np.uint64(0xF0), api=api)
    assert ((lhs[li.astype(int)] & np.uint64(0xF0)) == [0x20, 0x40, 0x60, 0x80, 0xA0]).all()
    lhs, rhs = u64([0, 0, 1]), u64([0, 0, 0, 0, 1])
    li, ri = ops.intersect(lhs, rhs, api=api)
    assert (lhs[li.astype(int)] == [0, 1]).all()
    lhs, rhs = u64([1, 5, 9]), u64([2, 5, 6, 7, 10])
    li, ri = ops.adjacent(lhs, rhs, api=api)
    assert (lhs[li.astype(int)] == [1, 5, 9]).all() and (rhs[ri.astype(int)] == [2, 6, 10]).all()
    assert (ops.merge(u64([1, 2, 5]), u64([2, 4]), api=api) == [1, 2, 2, 4, 5]).all()
    assert (ops.merge(u64([1, 2, 5]), u64([2, 4]), drop_duplicates=True, api=api) == [1, 2, 4, 5]).all()
    with pytest.raises(ValueError):
        ops.intersect(lhs, rhs, mask=np.uint64(0), api=api)
    assert len(ops.merge(u64([]), u64([]), api=api)) == 0
    assert len(ops.intersect(u64([]), u64([3]), api=api)[0]) == 0


@pytest.mark.parametrize("tag", ["128", "24179", "27685", "44358"])          # every complete triple the reference captured
def test_reference_fixture_arrays(api, tag):
    g = load_golden("snp_fixtures")
    lhs, rhs, mask = g[f"{tag}_lhs"], g[f"{tag}_rhs"], np.uint64(g[f"{tag}_mask"])
    li, ri = ops.intersect(lhs, rhs, mask=mask, api=api)
    assert np.array_equal(li, g[f"{tag}_int_drop_l"]) and np.array_equal(ri, g[f"{tag}_int_drop_r"])
    lk, rk = ops.intersect(lhs, rhs, mask=mask, drop_duplicates=False, api=api)
    assert np.array_equal(lk, g[f"{tag}_int_keep_l"]) and np.array_equal(rk, g[f"{tag}_int_keep_r"])
    got = ops.intersect_with_adjacents(lhs, rhs, mask=mask, api=api)
    for g_, key in zip(got, ("iwa_l", "iwa_r", "iwa_al", "iwa_ar")):
        assert np.array_equal(g_, g[f"{tag}_{key}"]), key
    al, ar = ops.adjacent(lhs, rhs, mask=mask, api=api)
    assert np.array_equal(al, g[f"{tag}_adj_l"]) and np.array_equal(ar, g[f"{tag}_adj_r"])
    assert np.array_equal(ops.merge(lhs, rhs, api=api), g[f"{tag}_merge"])
    assert np.array_equal(ops.merge(lhs, rhs, drop_duplicates=True, api=api), g[f"{tag}_merge_drop"])


@pytest.mark.parametrize("tag", ["128", "24179", "27685", "44358", "185", "45907", "90596"])
def test_reference_fixture_one_array_primitives(api, tag):
    """unique / popcount64_reduce (reference roaringish/unique.pyx, popcount.pyx) on every captured lhs array, the three captured
    without an rhs included; expected = the reference's outputs (tests/golden/make_golden.py)"""
    g = load_golden("snp_fixtures" if tag in ("128", "24179", "27685", "44358") else "snp_fixtures_lhs")
    lhs = g[f"{tag}_lhs"]
    assert np.array_equal(ops.unique(lhs, 36, api=api), g[f"{tag}_unique36"])
    k, c = ops.popcount64_reduce(lhs, 36, 0x3FFFF, api=api)
    assert np.array_equal(k, g[f"{tag}_pcr_keys"]) and np.array_equal(c, g[f"{tag}_pcr_counts"])
    if f"{tag}_unique18" in g.files:
        assert np.array_equal(ops.unique(lhs, 18, api=api), g[f"{tag}_unique18"])


def test_all_ones_header_quirk_of_the_drop_variants(api):
    """reference intersect.pyx:39,146,224-225: `last` starts as all ones, so a common value whose masked bits are all
    ones is dropped when it would be the first pair reported (it is then the only one) and reported after any other
    match.  Expected values are outputs of the reference itself (oracle/_ref) on these arrays; the oracle restates them."""
    mask = np.uint64(0xFFFFFFFFFFFC0000)
    top = 0xFFFFFFFFFFFC0000
    lone_l, lone_r = np.asarray([top | 7], dtype=np.uint64), np.asarray([3 << 18, top | 1], dtype=np.uint64)
    li, ri = ops.intersect(lone_l, lone_r, mask=mask, api=api)
    assert len(li) == 0 and len(ri) == 0
    got = ops.intersect_with_adjacents(lone_l, lone_r, mask=mask, api=api)
    assert all(len(x) == 0 for x in got)
    assert all(len(x) == 0 for x in O.intersect(lone_l, lone_r, mask=mask))
    lk, rk = ops.intersect(lone_l, lone_r, mask=mask, drop_duplicates=False, api=api)       # keep mode has no `last`
    assert np.array_equal(lk, [0]) and np.array_equal(rk, [1])
    lhs = np.asarray([1 << 18 | 3, 5 << 18 | 1, top | 7], dtype=np.uint64)
    rhs = np.asarray([1 << 18 | 1, 6 << 18 | 2, top | 1], dtype=np.uint64)
    li, ri = ops.intersect(lhs, rhs, mask=mask, api=api)
    assert np.array_equal(li, [0, 2]) and np.array_equal(ri, [0, 2])                         # reference: ([0, 2], [0, 2])
    al, ar = ops.adjacent(lhs, rhs, mask=mask, api=api)
    assert np.array_equal(al, [1]) and np.array_equal(ar, [1])                               # reference: ([1], [1])
    got = ops.intersect_with_adjacents(lhs, rhs, mask=mask, api=api)
    want = O.intersect_with_adjacents(lhs, rhs, mask=mask)
    for g_, w_ in zip(got, want):
        assert np.array_equal(g_, w_)
    assert np.array_equal(want[0], [0, 2]) and np.array_equal(want[2], [1])
    m8 = np.uint64(0xF0)
    l2, r2 = np.asarray([0x10, 0x25, 0xF3], dtype=np.uint64), np.asarray([0x00, 0x11, 0x30, 0xF1], dtype=np.uint64)
    li, ri = ops.intersect(l2, r2, mask=m8, api=api)
    assert np.array_equal(li, [0, 2]) and np.array_equal(ri, [1, 3])                         # reference: ([0, 2], [1, 3])
    li, ri = ops.intersect(l2[2:], r2, mask=m8, api=api)
    assert len(li) == 0
