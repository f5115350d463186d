"""RoaringishEncoder -- the wire format object of the index (reference roaringish/roaringish.py:54-282) --
against the reference's own outputs (tests/golden/encoder.npz, make_golden.py ONLY=encoder) for key widths
28, 32 and 20.  The set operations inside run through the C ABI (emulated kernels here, gfx950 with -m gpu)."""
import numpy as np
import pytest

from searcharray_amd.roaringish import (RoaringishEncoder, convert_keys, n_msb_mask, DEFAULT_KEY_MASK,
                                        DEFAULT_PAYLOAD_LSB_MASK, DEFAULT_PAYLOAD_MSB_MASK)
from tests.helpers import load_golden


def test_default_layout_constants():
    enc = RoaringishEncoder()
    assert enc.key_mask == DEFAULT_KEY_MASK and enc.payload_msb_mask == DEFAULT_PAYLOAD_MSB_MASK
    assert enc.payload_lsb_mask == DEFAULT_PAYLOAD_LSB_MASK
    assert enc.header_mask == np.uint64(0xFFFFFFFFFFFC0000) and enc.max_payload == 2 ** 18 - 1
    assert int(enc.payload_lsb_bits) == 18 and int(enc.payload_msb_bits) == 18 and int(enc.header_bits) == 46
    assert n_msb_mask(np.uint64(4)) == np.uint64(0xF000000000000000)
    with pytest.raises(ValueError, match="Positions must be less than 262144"):
        enc.validate_payload(np.asarray([2 ** 18], dtype=np.uint64))


@pytest.mark.parametrize("kb", [28, 32, 20])
def test_encoder_matches_reference(default_api, kb):
    g = load_golden("encoder")
    t = f"k{kb}_"
    enc = RoaringishEncoder(np.uint64(kb))
    L = int(enc.payload_lsb_bits)
    words, nb = enc.encode(keys=g[t + "keys"], payload=g[t + "posns"], boundaries=g[t + "bounds"])
    assert words.dtype == np.uint64 and np.array_equal(words, g[t + "enc_b"])
    assert nb.dtype == np.uint64 and np.array_equal(nb, g[t + "enc_b_bounds"])
    a, b = words[int(nb[0]):int(nb[1])], words[int(nb[1]):int(nb[2])]
    n0 = int(g[t + "bounds"][1])
    single, none = enc.encode(keys=g[t + "keys"][:n0], payload=g[t + "posns"][:n0])
    assert none is None and np.array_equal(single, a)
    n1 = int(g[t + "bounds"][2])
    assert np.array_equal(enc.encode(payload=g[t + "posns"][n0:n1][:12])[0], g[t + "enc_nokeys"])
    dec = enc.decode(a)
    assert np.array_equal(np.asarray([k for k, _ in dec], dtype=np.uint64), g[t + "dec_keys"])
    assert np.array_equal([len(v) for _, v in dec], g[t + "dec_lens"])
    assert np.array_equal(np.concatenate([v for _, v in dec]), g[t + "dec_vals"])
    assert len(enc.decode(a, get_keys=False)) == int(g[t + "dec_nokeys_n"])
    # encode(decode(x)) == x
    rk = np.concatenate([np.full(len(v), k, dtype=np.uint64) for k, v in dec])
    assert np.array_equal(enc.encode(keys=rk, payload=np.concatenate([v for _, v in dec]))[0], a)
    k, c = enc.num_values_per_key(a)
    assert np.array_equal(k, g[t + "nvpk_k"]) and np.array_equal(c, g[t + "nvpk_c"]) and c.dtype == g[t + "nvpk_c"].dtype
    assert np.array_equal(enc.keys(a), g[t + "keys_of"]) and np.array_equal(enc.keys_unique(a), g[t + "keys_unique"])
    assert np.array_equal(enc.payload_msb(a), g[t + "msb"]) and np.array_equal(enc.payload_lsb(a), g[t + "lsb"])
    assert np.array_equal(enc.header(a), g[t + "hdr"])
    for name, res in (("cand", enc.intersect_candidates(a, b)), ("rshift", enc.intersect_rshift(a, b)),
                      ("isect", enc.intersect(a, b))):
        for j, r in enumerate(res):
            assert np.array_equal(r, g[f"{t}{name}_{j}"]), (name, j)
    assert np.array_equal(enc.key_partition(a, np.uint64(200), 2), g[t + "part2"])
    assert np.array_equal(enc.key_partition(a, np.uint64(200), 8), g[t + "part8"])
    some = g[t + "slice_keys_in"]
    assert np.array_equal(enc.slice(a, keys=some), g[t + "slice_keys"])
    assert np.array_equal(enc.slice(a, header=enc.header(b)), g[t + "slice_hdr"])
    assert np.array_equal(enc.slice(a, min_payload=L, max_payload=3 * L - 1), g[t + "slice_posn"])
    assert np.array_equal(enc.slice(a, keys=some, max_payload=2 * L - 1), g[t + "slice_keys_posn"])
    with pytest.raises(ValueError, match="Can't specify both"):
        enc.slice(a, keys=some, header=enc.header(b))
    with pytest.raises(ValueError, match=f"multiple of {L}"):
        enc.slice(a, min_payload=1)
    with pytest.raises(ValueError, match=f"multiple of {L} - 1"):
        enc.slice(a, max_payload=L)


def test_convert_keys_and_empty():
    g = load_golden("encoder")
    got = np.concatenate([convert_keys(5), convert_keys([3, 1]), convert_keys(range(2, 6)), convert_keys(range(0))])
    assert got.dtype == np.uint64 and np.array_equal(got, g["convert"])
    with pytest.raises(ValueError, match="Unknown type"):
        convert_keys("x")
    enc = RoaringishEncoder()
    w, nb = enc.encode(payload=np.empty(0, np.uint64), keys=np.empty(0, np.uint64))
    assert len(w) == 0 and nb is None
    assert enc.decode(np.empty(0, np.uint64)) == []


def test_encoder_agrees_with_the_index_builder():
    """The module's fast triple encoder (index build) and the encoder object produce the same words."""
    from searcharray_amd import roaringish as rz
    from tests.helpers import golden_corpus
    g, (t, d, p), lens, num_docs, vocab = golden_corpus("zipf_small")
    words, wt = rz.encode_sorted(t, d, p)
    bounds = np.flatnonzero(np.concatenate([[True], t[1:] != t[:-1]])).astype(np.uint64)
    w2, nb = RoaringishEncoder().encode(keys=d, payload=p, boundaries=bounds)
    assert np.array_equal(w2, words)
    present = np.unique(t)
    off = rz.term_offsets(wt, vocab)
    assert np.array_equal(nb[:-1], off[present]) and int(nb[-1]) == len(words)
