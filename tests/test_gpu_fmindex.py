"""FM-index build + count (+ locate) on the device against the oracle's restatement of
countFMIndex / locateFMIndex (incl. Q10), and against naive substring counting."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import textcomp
    c = textcomp.Context(0)
    yield c
    c.close()


def test_doc_example_abracadabra(ctx, golden):
    d = golden["fmindex_doc"]
    fm = ctx.fm_build(d["text"].encode())
    info = fm.info()
    got_c = {("$" if s < 0 else chr(s)): int(v) for s, v in zip(info["c_sym"], info["c_val"])}
    assert got_c == d["C"] and info["N"] == 12 and info["primary"] == d["L"].index("$")
    pats = [b"abra", b"a", b"abracadabra", b"x", b"xra", b"rab", b"", b"bra", b"cad", b"abracadabrax"]
    got = fm.count(pats)
    ofm = O.FMIndex(d["text"].encode())
    assert [int(v) or None for v in got] == [ofm.count(p) for p in pats]
    assert [int(v) for v in got[:5]] == [2, 5, 1, 0, 2]                    # Q10: "xra" -> Just 2
    loc = fm.locate(pats)
    assert [h.tolist() for h in loc] == [ofm.locate(p) for p in pats]
    fm.close()


def test_count_matches_oracle_random(ctx):
    rng = np.random.default_rng(21)
    for sigma_bytes, n in [(b"ACGTN", 5000), (b"ab", 3000), (bytes(range(256)), 20000), (b"A", 700),
                           (b"ACGTN", 1 << 20)]:
        t = bytes(rng.choice(list(sigma_bytes), n).astype(np.uint8))
        fm = ctx.fm_build(t)
        ofm = O.FMIndex(t)
        pats = []
        for _ in range(400):
            m = int(rng.integers(1, 24))
            if rng.random() < 0.7:
                o = int(rng.integers(0, max(1, n - m)))
                pats.append(t[o:o + m])
            else:
                pats.append(bytes(rng.integers(0, 256, m, dtype=np.uint8)) if rng.random() < 0.3
                            else bytes(rng.choice(list(sigma_bytes), m).astype(np.uint8)))
        got = fm.count(pats)
        assert [int(v) or None for v in got] == [ofm.count(p) for p in pats]
        some = pats[:60]
        assert [h.tolist() for h in fm.locate(some)] == [ofm.locate(p) for p in some]
        fm.close()


def test_count_vs_naive_substring(ctx):
    rng = np.random.default_rng(8)
    t = bytes(rng.choice(list(b"ACGTN"), 4000).astype(np.uint8))
    fm = ctx.fm_build(t)
    pats = [bytes(rng.choice(list(b"ACGTN"), int(rng.integers(1, 7))).astype(np.uint8)) for _ in range(300)]
    got = fm.count(pats)
    for p, c in zip(pats, got):
        assert int(c) == sum(1 for i in range(len(t) - len(p) + 1) if t[i:i + len(p)] == p)
    fm.close()


def test_mirror_count_locate_shapes(ctx):
    from textcomp import fmindex
    assert fmindex.bytestringFMIndexCountS([], b"abc") == []          # FMIndex.hs:365
    assert fmindex.bytestringFMIndexCountS([b"a"], b"") == []         # FMIndex.hs:366
    got = fmindex.bytestringFMIndexCountS([b"abra", b"zz", b""], b"abracadabra")
    assert got == [(b"abra", 2), (b"zz", None), (b"", None)] == O.bytestringFMIndexCountS([b"abra", b"zz", b""], b"abracadabra")
    assert fmindex.bytestringFMIndexCountP([b"abra"], b"abracadabra") == [(b"abra", 2)]
    loc = fmindex.bytestringFMIndexLocateS([b"abra", b"zz"], b"abracadabra")
    assert loc[0][0] == b"abra" and sorted(loc[0][1]) == [1, 8] and loc[1] == (b"zz", [])


def test_config4_shape_small(ctx):
    """BASELINE configs[3] at a parity-checkable size: 100-byte ACGTN patterns, 99 % substrings,
    1 % iid (miss path), against the oracle."""
    n, npat = 1 << 20, 5000
    t = O.gen_acgtn(0xC4, n).tobytes()
    rng = np.random.default_rng(0xC4F0)
    pats = []
    for j in range(npat):
        if j % 100 == 99:
            pats.append(O.gen_acgtn(0xC4F000 + j, 100).tobytes())
        else:
            o = int(rng.integers(0, n - 99))
            pats.append(t[o:o + 100])
    fm = ctx.fm_build(t)
    got = fm.count(pats)
    ofm = O.FMIndex(t)
    assert [int(v) or None for v in got] == [ofm.count(p) for p in pats]
    assert int((got == 0).sum()) >= npat // 100 - 1
    fm.close()


def test_fm_randomized_against_oracle_and_naive(ctx):
    """Random texts (lengths around multiples of 64 included: the Occ checkpoints of both sides), random
    patterns -- substrings, random strings, strings with a foreign byte, the empty pattern: count and
    locate equal the oracle's; count equals naive matching whenever every pattern byte occurs."""
    rng = np.random.default_rng(4242)
    for it in range(40):
        n = int(rng.choice([63, 64, 127, 128, 639, 4095])) + int(rng.integers(0, 2)) if it % 2 else int(rng.integers(1, 20000))
        sigma = int(rng.integers(1, 8)) if it % 3 else int(rng.integers(8, 257))
        alpha = rng.permutation(256)[:sigma]
        t = alpha[rng.integers(0, sigma, n)].astype(np.uint8)
        if n > 8:
            ln = int(rng.integers(1, n // 2)); a0, b0 = int(rng.integers(0, n - ln)), int(rng.integers(0, n - ln))
            t[b0:b0 + ln] = t[a0:a0 + ln].copy()
        tb = t.tobytes()
        pats = [tb[-1:], tb[:1], bytes([int(alpha.max())]), b""]
        for _ in range(25):
            m = int(rng.integers(1, 30)); a0 = int(rng.integers(0, n))
            pats.append(tb[a0:a0 + m])
            pats.append(alpha[rng.integers(0, sigma, m)].astype(np.uint8).tobytes())
            p = bytearray(tb[a0:a0 + m]); p[int(rng.integers(0, len(p)))] = int(rng.integers(0, 256)); pats.append(bytes(p))
        ofm = O.FMIndex(tb)
        fm = ctx.fm_build(tb)
        counts, hits = fm.count(pats), fm.locate(pats)
        fm.close()
        for p, c, h in zip(pats, counts, hits):
            assert (None if c == 0 else int(c)) == ofm.count(p), (it, n, p)
            assert sorted(int(v) for v in h) == sorted(ofm.locate(p)), (it, n, p)
            if p and all(bytes([b]) in tb for b in p):
                assert int(c) == sum(1 for i in range(n - len(p) + 1) if tb.startswith(p, i)), (it, n, p)
