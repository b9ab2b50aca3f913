"""Oracle self-consistency and the reference's quirks. CPU only."""
import numpy as np
import pytest

import oracle as O


def _rand_texts():
    rng = np.random.default_rng(7)
    out = [b"", b"a", b"ba", b"ab", b"aaaaaaaa", b"abababab", b"mississippi", b"ACGTACGTACGT",
           bytes(range(256)), bytes([255, 0, 255, 0, 0])]
    for sigma in (1, 2, 5, 256):
        for n in (1, 2, 3, 17, 257, 3000):
            out.append(rng.integers(0, sigma, n, dtype=np.uint8).tobytes())
    return out


@pytest.mark.parametrize("t", _rand_texts(), ids=lambda t: "n%d" % len(t))
def test_sa_doubling_equals_naive(t):
    assert O.suffix_array(t).tolist() == O.suffix_array(t, naive=True).tolist()


def test_sa_doubling_adversarial():
    for t in (b"A" * 5000, b"ACGT" * 2000, b"N" * 3000 + b"A" + b"N" * 3000):
        assert O.suffix_array(t).tolist() == O.suffix_array(t, naive=True).tolist()


def test_bwt_shape_q1_q2():
    assert O.bytestringToBWT(b"") == []                      # BWT.hs:58
    L = O.bytestringToBWT(b"banana")
    assert L.count(None) == 1 and L[0] == ord("a") and len(L) == 7   # Q1
    assert "".join("$" if v is None else chr(v) for v in L) == "annb$aa"


@pytest.mark.parametrize("t", _rand_texts(), ids=lambda t: "n%d" % len(t))
def test_bwt_roundtrip(t):
    assert O.bytestringFromWord8BWT(O.bytestringToBWT(t)) == t


def test_inverse_bwt_generic_q9():
    assert O.bytestringFromWord8BWT([ord("a"), ord("b")]) == b""      # no Nothing => empty
    assert O.bytestringFromWord8BWT([None]) == b""
    # two Nothings (what the Q6 RLE round trip produces for "ba"): walk starts at the first
    assert O.bytestringFromWord8BWT([ord("a"), ord("b"), None, None]) == b"a"


def test_rle_quirks_q5_q6_q7():
    a, b = ord("a"), ord("b")
    r = lambda x: O.render_rle(*O.rle_encode_arr(O.arr_of(x)))
    assert r([]) == []
    # Q5: runs never merge across the sentinel
    assert r([a, a, None, a]) == [b"2", b"a", b"1", None, b"1", b"a"]
    # Q6: trailing sentinel => extra (stale count, Nothing)
    assert r([a, b, None]) == [b"1", b"a", b"1", b"b", b"1", None, b"1", None]
    assert r([a, a, a, None]) == [b"3", b"a", b"1", None, b"3", None]
    assert O.bytestringToBWTToRLEB(b"ba") == [b"1", b"a", b"1", b"b", b"1", None, b"1", None]
    # Q7: leading Nothing followed by a Just is dropped; [Nothing] alone; consecutive Nothings
    assert r([None, a]) == [b"1", b"a"]
    assert r([None]) == [b"1", None]
    assert r([None, None]) == [b"1", None, b"1", None, b"1", None]
    assert r([a, None, None, b]) == [b"1", b"a", b"1", None, b"1", None, b"1", None, b"1", b"b"]


def test_rle_decode_q8():
    a = ord("a")
    assert O.bytestringBWTFromRLEB([b"3", b"a", b"7", None, b"1", b"b"]) == [a, a, a, None, ord("b")]
    assert O.bytestringBWTFromRLEB([b"2", b"a", b"9"]) == [a, a]        # odd tail ignored
    assert O.bytestringBWTFromRLEB([b"0", b"a"]) == []
    with pytest.raises(O.OracleMalformed):
        O.bytestringBWTFromRLEB([b"x", b"a"])
    with pytest.raises(O.OracleMalformed):
        O.bytestringBWTFromRLEB([None, b"a"])
    # Q6: the reference's own round trip is broken for texts that are their own greatest suffix
    assert O.bytestringFromBWTFromRLEB(O.bytestringToBWTToRLEB(b"ba")) == b"a"


def test_mtf_q3_q4():
    idx, fl = O.bytestringBWTToMTFB([ord("c"), ord("a"), None, ord("c")])
    # alphabet = sorted present symbols, Nothing first: [$, a, c]
    assert idx == [2, 2, 2, 2] and fl == [b"c", None, b"a"]
    assert O.bytestringBWTToMTFB([]) == ([], [])
    assert O.bytestringBWTFromMTFB(([2, 2, 2, 2], [b"c", None, b"a"])) == [ord("c"), ord("a"), None, ord("c")]
    assert O.bytestringBWTFromMTFB(([], [b"a"])) == [] and O.bytestringBWTFromMTFB(([0], [])) == []
    with pytest.raises(O.OracleMalformed):
        O.bytestringBWTFromMTFB(([3], [b"c", None, b"a"]))


@pytest.mark.parametrize("t", _rand_texts(), ids=lambda t: "n%d" % len(t))
def test_mtf_rle_roundtrips(t):
    bwt = O.bytestringToBWT(t)
    assert O.bytestringBWTFromMTFB(O.bytestringBWTToMTFB(bwt)) == bwt
    rle = O.bytestringBWTToRLEB(bwt)
    back = O.bytestringBWTFromRLEB(rle)
    if bwt and bwt[-1] is None:       # Q6 class
        assert back == bwt + [None]
    else:
        assert back == bwt


def test_count_vs_naive():
    rng = np.random.default_rng(3)
    for _ in range(60):
        n = int(rng.integers(1, 400))
        t = bytes(rng.choice(list(b"ACGTN"), n).astype(np.uint8))
        fm = O.FMIndex(t)
        for _ in range(30):
            m = int(rng.integers(1, 8))
            p = bytes(rng.choice(list(set(t)), m).astype(np.uint8))    # every pattern byte occurs in the text
            naive = sum(1 for i in range(n - m + 1) if t[i:i + m] == p)
            assert fm.count(p) == (naive or None)
            assert sorted(fm.locate(p)) == [i + 1 for i in range(n - m + 1) if t[i:i + m] == p]


def test_generators_deterministic():
    a = O.gen_acgtn(0xC2, 1000)
    assert set(a.tolist()) <= set(b"ACGTN") and a.tolist() == O.gen_acgtn(0xC2, 1000).tolist()
    assert O.gen_acgtn(0xC2, 2000)[:1000].tolist() == a.tolist()      # counter based
    b = O.gen_ascii(0xC1, 1000)
    assert b.min() >= 0x20 and b.max() <= 0x7e


def test_fm_count_at_checkpoint_multiples():
    """Occ(c, N) with N a multiple of the oracle's checkpoint distance (a pattern ending in the largest
    symbol starts from e = N): found by tests/long/fuzz_fm.py -- counts equal naive matching when every
    pattern byte occurs in the text."""
    rng = np.random.default_rng(64)
    for n in (63, 127, 128, 191, 639, 640, 27967):
        t = bytes(rng.choice(list(b"ioz\xcf"), n).astype(np.uint8))
        fm = O.FMIndex(t)
        for p in (b"\xcf", b"o\xcf", b"oo\xcf", b"z", b"iz", t[3:9], t[-4:], t[-1:]):
            naive = sum(1 for i in range(n - len(p) + 1) if t.startswith(p, i))
            assert fm.count(p) == (naive or None), (n, p)
            assert sorted(fm.locate(p)) == [i + 1 for i in range(n - len(p) + 1) if t.startswith(p, i)], (n, p)
