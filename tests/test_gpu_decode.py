"""Parity of the HIP decode path (inverse RLE / MTF / BWT, through the C ABI) against
the CPU oracle, incl. the reference's behaviour on malformed sequences (Q8, Q9)."""
import numpy as np
import pytest

import oracle as O
from test_gpu_encode import TEXTS, _ids

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import textcomp
    c = textcomp.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("t", TEXTS, ids=_ids)
def test_bwt_decode_roundtrip(ctx, t):
    L = O.bwt_encode_arr(t)
    prim = int(np.nonzero(L < 0)[0][0])
    Lb = np.where(L < 0, 0, L).astype(np.uint8)
    assert ctx.bwt_decode(Lb, prim) == bytes(t)
    assert ctx.bwt_decode_sym(L) == bytes(t)


def test_bwt_decode_generic_q9(ctx):
    import textcomp
    a, b = ord("a"), ord("b")
    assert ctx.bwt_decode_sym(np.array([a, b], dtype=np.int16)) == b""          # no Nothing
    assert ctx.bwt_decode_sym(np.array([-1], dtype=np.int16)) == b""
    assert ctx.bwt_decode_sym(np.array([a, b, -1, -1], dtype=np.int16)) == b"a"  # Q6 round trip of "ba"
    rng = np.random.default_rng(9)
    n_mal = 0
    for _ in range(200):
        n = int(rng.integers(1, 600))
        x = rng.integers(0, 4, n).astype(np.int16)
        for _ in range(int(rng.integers(0, 3))):
            x[int(rng.integers(0, n))] = -1
        try:
            exp = O.bwt_decode_arr(x)
        except O.OracleMalformed:
            n_mal += 1
            with pytest.raises(textcomp.TcMalformed):
                ctx.bwt_decode_sym(x)
            continue
        assert ctx.bwt_decode_sym(x) == exp
    assert n_mal > 0


@pytest.mark.parametrize("t", TEXTS, ids=_ids)
def test_mtf_decode(ctx, t):
    L = O.bwt_encode_arr(t)
    idx, fl = O.mtf_encode_arr(L)
    out = ctx.mtf_decode(idx.astype(np.uint16), fl)
    assert out.tolist() == L.tolist()


def test_mtf_decode_edges(ctx):
    import textcomp
    # initial list = sort(unique(list)) even when the given list has duplicates (:214)
    out = ctx.mtf_decode(np.array([2, 2, 2, 2], np.uint16), np.array([99, -1, 97, 99], np.int16))
    assert out.tolist() == O.mtf_decode_arr([2, 2, 2, 2], [99, -1, 97, 99]).tolist() == [99, 97, -1, 99]
    assert len(ctx.mtf_decode(np.array([], np.uint16), np.array([97], np.int16))) == 0
    assert len(ctx.mtf_decode(np.array([0], np.uint16), np.array([], np.int16))) == 0
    with pytest.raises(textcomp.TcMalformed):
        ctx.mtf_decode(np.array([0, 3, 0], np.uint16), np.array([99, -1, 97], np.int16))


def test_rle_decode_q8(ctx):
    rng = np.random.default_rng(4)
    cases = [([3, 7, 1], [97, -1, 98]), ([0, 2], [97, 98]), ([5], [-1]), ([1000, 1, 70000], [1, 2, 3])]
    for _ in range(40):
        k = int(rng.integers(1, 3000))
        cases.append((rng.integers(0, 50, k).tolist(), rng.integers(-1, 4, k).tolist()))
    for counts, syms in cases:
        exp = O.rle_decode_arr(counts, syms)
        got = ctx.rle_decode(np.array(counts, np.uint32), np.array(syms, np.int16))
        assert got.tolist() == exp.tolist()
    got = ctx.rle_decode_u16(np.array([2, 3], np.uint32), np.array([65535, 7], np.uint16))
    assert got.tolist() == [65535, 65535, 7, 7, 7]


@pytest.mark.parametrize("t", TEXTS, ids=_ids)
def test_fused_roundtrip(ctx, t):
    blk = ctx.encode(t)
    assert ctx.decode(blk) == bytes(t)


def test_rle_decode_long_runs(ctx):
    """Runs below / at / above the wave-fill (32) and grid-fill (16384) thresholds, mixed with
    Nothing pairs; and the inverse BWT of periodic texts (walks with a constant row mod 256 must
    still meet splitters)."""
    counts = np.array([1, 31, 32, 33, 16383, 7, 16384, 16385, 1, 300000, 2, 5], dtype=np.uint32)
    syms = np.array([5, 6, -1, 7, 8, -1, 9, 10, 11, 12, 13, -1], dtype=np.int16)
    exp = np.concatenate([np.full(1 if s < 0 else c, s, np.int16) for c, s in zip(counts, syms)])
    assert np.array_equal(ctx.rle_decode(counts, syms), exp)
    vals = np.array([5, 6, 0, 7, 8, 65535, 9, 10, 11, 12, 13, 1], dtype=np.uint16)
    exp = np.concatenate([np.full(c, v, np.uint16) for c, v in zip(counts, vals)])
    assert np.array_equal(ctx.rle_decode_u16(counts, vals), exp)
    rng = np.random.default_rng(77)
    for block, copies in ((4096, 256), (1000, 300), (256, 1024)):
        t = (bytes(rng.choice(list(b"ACGT"), block).astype(np.uint8)) * copies)
        assert ctx.decode(ctx.encode(t)) == t
    t = b"A" * 200000
    assert ctx.decode(ctx.encode(t)) == t


@pytest.mark.parametrize("env", [{"TC_IBWT_SCATTER": "0"}, {"TC_IBWT_REWALK": "1"},
                                 {"TC_IBWT_SCATTER": "0", "TC_IBWT_REWALK": "1"}, {"TC_IBWT_LF": "0"},
                                 {"TC_IBWT_LF": "0", "TC_IBWT_REWALK": "1"},
                                 {"TC_IBWT_LF": "0", "TC_IBWT_SCATTER": "0"}, {"TC_MTF_FORCE_GENERAL": "1"},
                                 {"TC_DECODE_BYTES": "0"}, {"TC_DECODE_BYTES": "0", "TC_IBWT_LF": "0"},
                                 {"TC_IBWT_SEGCAP": "1024"}, {"TC_IBWT_SEGCAP": "1024", "TC_IBWT_LF": "0"},
                                 {"TC_IBWT_SEGCAP": "64"}, {"TC_IBWT_SEGCAP": "64", "TC_IBWT_LF": "0"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_ibwt_path_selectors(ctx, env, monkeypatch):
    """The alternatives inside the inverse BWT (generic radix sort of the positions instead of the
    dedicated scatter; every recorded segment treated as overflowed and walked again; small alphabets
    by positions instead of by LF over the packed last column; the inverse MTF of a small alphabet on the
    byte-list kernels instead of the nibble ones; records that count as full early, so that walks continue in extra
    splitter slots -- 1024 -- or use them up and are repeated the old way -- 64) decode to the same text."""
    rng = np.random.default_rng(5)
    texts = [O.gen_acgtn(31, 1 << 20).tobytes(), O.gen_ascii(32, 200000).tobytes(),
             bytes(rng.integers(0, 256, 100000, dtype=np.uint8)),
             bytes(rng.choice(list(b"ACGT"), 1000).astype(np.uint8)) * 300,
             O.gen_acgtn(33, 127).tobytes(), b"A" * 5000 + b"C", b"AC" * 40000, b"G"]
    blks = [ctx.encode(t) for t in texts]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for t, blk in zip(texts, blks):
        assert ctx.decode(blk) == t


def test_fused_decode_rejects_inconsistent_block(ctx):
    import textcomp
    blk = ctx.encode(b"mississippi" * 50)
    bad = dict(blk)
    bad["run_count"] = blk["run_count"].copy()
    bad["run_count"][3] += 1
    with pytest.raises(textcomp.TcError):
        ctx.decode(bad)


@pytest.mark.parametrize("gen,seed,n", [("ascii", 0xC1, 65536), ("acgtn", 0xC2, 1 << 22)])
def test_config_roundtrip(ctx, gen, seed, n):
    t = (O.gen_ascii if gen == "ascii" else O.gen_acgtn)(seed, n)
    blk = ctx.encode(t)
    assert ctx.decode(blk) == t.tobytes()
