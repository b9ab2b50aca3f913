"""Parity of the HIP encode path (through the C ABI) against the CPU oracle.
Bit-exact: everything here is integer / byte / index work."""
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


def _texts():
    rng = np.random.default_rng(11)
    out = [b"a", b"ba", b"ab", b"aaaaaaaa", b"abababab", b"mississippi", b"abracadabra",
           b"ACGTACGTACGT", bytes(range(256)), bytes([255, 0, 255, 0, 0]), b"aaaabbbbcccc"]
    for sigma in (1, 2, 5, 17, 100, 256):
        for n in (1, 2, 3, 63, 64, 65, 257, 4095, 4096, 4097, 20000):
            out.append(rng.integers(0, sigma, n, dtype=np.uint8).tobytes())
    return out


TEXTS = _texts()


def _ids(t):
    return "n%d_s%d" % (len(t), len(set(t)))


def _expect_bwt(t):
    L = O.bwt_encode_arr(t)
    prim = int(np.nonzero(L < 0)[0][0])
    Lb = L.copy()
    Lb[prim] = 0
    return Lb.astype(np.uint8), prim, L


@pytest.mark.parametrize("t", TEXTS, ids=_ids)
def test_suffix_array_and_bwt(ctx, t):
    sa = ctx.suffix_array(t)
    assert sa.tolist() == O.suffix_array(t).tolist()
    L, prim = ctx.bwt_encode(t)
    eL, eprim, _ = _expect_bwt(t)
    assert prim == eprim and L.tolist() == eL.tolist()


def test_bwt_empty(ctx):
    L, prim = ctx.bwt_encode(b"")
    assert len(L) == 0 and prim is None


@pytest.mark.parametrize("t", TEXTS, ids=_ids)
def test_mtf(ctx, t):
    eL, eprim, sym = _expect_bwt(t)
    idx, fl = ctx.mtf_encode(eL, eprim)
    eidx, efl = O.mtf_encode_arr(sym)
    assert idx.tolist() == eidx.tolist() and fl.tolist() == efl.tolist()
    idx2, fl2 = ctx.mtf_encode_sym(sym)
    assert idx2.tolist() == eidx.tolist() and fl2.tolist() == efl.tolist()


def test_mtf_general_path_equals_nibble_path(ctx, monkeypatch):
    t = O.gen_acgtn(5, 50000)
    eL, eprim, sym = _expect_bwt(t)
    a = ctx.mtf_encode(eL, eprim)
    monkeypatch.setenv("TC_MTF_FORCE_GENERAL", "1")
    b = ctx.mtf_encode(eL, eprim)
    assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()
    assert a[0].tolist() == O.mtf_encode_arr(sym)[0].tolist()


@pytest.mark.parametrize("sigma", [17, 40, 64, 65, 100, 128, 129, 192, 193, 255, 256, 257])
def test_mtf_general_sigma(ctx, sigma, monkeypatch):
    """sigma > 16: one chunk per lane (byte lists in LDS, sigma <= 256) or per wave (sigma = 257);
    lengths around the 128-symbol lane chunk and the 32768-symbol tile; skewed data (small ranks,
    late first occurrences) and uniform data (large ranks)."""
    rng = np.random.default_rng(1000 + sigma)
    nsym = sigma - 1 if sigma == 257 or sigma % 2 else sigma       # with / without a Nothing
    with_nothing = nsym < sigma
    alphabet = rng.permutation(256)[:nsym]
    for N, skew in ((1, 0), (127, 1), (129, 0), (32768, 1), (32769, 0), (100001, 1), (70000, 0),
                    (1100000, 0), (1100001, 1)):     # >= 2^20: the path is chosen by a sample of the stream
        if skew:
            p = 1.0 / np.arange(1, nsym + 1) ** 1.5
            body = rng.choice(nsym, N, p=p / p.sum())
            late = rng.integers(0, N, min(N, nsym))        # every symbol occurs at least once, somewhere
            body[late[:min(N, nsym)]] = np.arange(min(N, nsym))
        else:
            body = rng.integers(0, nsym, N)
            if N >= nsym:
                body[rng.permutation(N)[:nsym]] = np.arange(nsym)
        sym = alphabet[body].astype(np.int16)
        if with_nothing and N > 1:
            sym[rng.integers(0, N)] = -1
        eidx, efl = O.mtf_encode_arr(sym)
        monkeypatch.setenv("TC_MTF_TS", "2")                       # timestamps (the default beyond 64 symbols)
        idx, fl = ctx.mtf_encode_sym(sym)
        assert np.array_equal(idx, eidx) and fl.tolist() == efl.tolist(), (sigma, N, skew)
        monkeypatch.setenv("TC_MTF_TS", "0")                       # the list-shifting families
        idx, fl = ctx.mtf_encode_sym(sym)
        assert np.array_equal(idx, eidx) and fl.tolist() == efl.tolist(), (sigma, N, skew)
        if N == 100001 or N == 1100000:
            monkeypatch.setenv("TC_MTF_WAVE_CHUNKS" if N == 100001 else "TC_MTF_RANK_SAMPLE", "1" if N == 100001 else "0")
            idx, fl = ctx.mtf_encode_sym(sym)
            monkeypatch.delenv("TC_MTF_WAVE_CHUNKS" if N == 100001 else "TC_MTF_RANK_SAMPLE")
            assert np.array_equal(idx, eidx) and fl.tolist() == efl.tolist()
        if (sym >= 0).all() or int(np.sum(sym < 0)) == 1:
            # (L, primary) accessor form: 16-byte staged loads + sentinel patch
            prim = int(np.nonzero(sym < 0)[0][0]) if (sym < 0).any() else None
            Lb = np.where(sym < 0, 0, sym).astype(np.uint8)
            idx, fl = ctx.mtf_encode(Lb, prim)
            assert np.array_equal(idx, eidx) and fl.tolist() == efl.tolist()
            monkeypatch.setenv("TC_MTF_TS", "2")
            idx, fl = ctx.mtf_encode(Lb, prim)
            assert np.array_equal(idx, eidx) and fl.tolist() == efl.tolist()
        monkeypatch.delenv("TC_MTF_TS", raising=False)
        dec = ctx.mtf_decode(eidx, efl)
        assert np.array_equal(dec, sym)


def _texts_257():
    rng = np.random.default_rng(257)
    out = []
    for n in (300, 5000, 40000, 1200000):
        t = rng.integers(0, 256, n).astype(np.uint8)
        t[rng.permutation(n)[:256]] = np.arange(256, dtype=np.uint8)
        out.append(t.tobytes())
    # skewed: few values dominate, the others turn up one by one, far apart
    n = 1300000
    t = rng.choice(np.array([65, 66, 67, 200], np.uint8), n, p=[0.5, 0.3, 0.15, 0.05])
    pos = np.sort(rng.permutation(n)[:256])
    t[pos] = rng.permutation(256).astype(np.uint8)
    out.append(t.tobytes())
    out.append(bytes(range(256)) * 700)                      # periodic, every value
    out.append(bytes(range(255, -1, -1)) + b"\x00" * 70000)   # sentinel row near one end of the BWT
    out.append(b"\xff" * 70000 + bytes(range(256)))
    return out


@pytest.mark.parametrize("t", _texts_257(), ids=lambda t: "n%d" % len(t))
def test_sigma_257_sentinel_split(ctx, t, monkeypatch):
    """All 256 byte values + the sentinel: the 256-symbol lane chunks with their fix-ups (encode:
    +1 at first occurrences; decode: the chain of rows whose rank exceeds the number of values met)
    give the oracle's block, decode it, and agree with the nine-bit path; a block whose primary does
    not match its index stream decodes as on the nine-bit path."""
    _, eprim, sym = _expect_bwt(t)
    eidx, efl = O.mtf_encode_arr(sym)
    ec, ev = O.rle_encode_u32_arr(eidx)
    blk = ctx.encode(t)
    assert blk["sigma"] == 257 and blk["primary"] == eprim and blk["final_list"].tolist() == efl.tolist()
    assert np.array_equal(blk["run_count"], ec) and np.array_equal(blk["run_value"], ev)
    assert ctx.decode(blk) == t
    bad = dict(blk)
    bad["primary"] = blk["primary"] - 1 if blk["primary"] > 1 else blk["primary"] + 1

    def outcome(b):
        import textcomp
        try:
            return ctx.decode(b)
        except textcomp.TcError as e:
            return ("error", e.args[0])
    got_bad = outcome(bad)
    monkeypatch.setenv("TC_MTF_TS", "0")                 # the list-shifting families: sampled choice,
    blk2 = ctx.encode(t)
    assert blk2["final_list"].tolist() == efl.tolist() and np.array_equal(blk2["run_count"], ec)
    assert np.array_equal(blk2["run_value"], ev)
    monkeypatch.setenv("TC_MTF_RANK_SAMPLE", "0")        # lane chunks whatever the ranks look like
    blk1 = ctx.encode(t)
    assert blk1["final_list"].tolist() == efl.tolist() and np.array_equal(blk1["run_count"], ec)
    assert np.array_equal(blk1["run_value"], ev)
    monkeypatch.setenv("TC_MTF_SENTINEL_SPLIT", "0")
    blk0 = ctx.encode(t)
    assert blk0["final_list"].tolist() == efl.tolist() and np.array_equal(blk0["run_count"], ec)
    assert np.array_equal(blk0["run_value"], ev)
    assert ctx.decode(blk) == t
    assert outcome(bad) == got_bad


@pytest.mark.parametrize("t", TEXTS, ids=_ids)
def test_rle(ctx, t):
    eL, eprim, sym = _expect_bwt(t)
    counts, syms = ctx.rle_encode(eL, eprim)
    ec, es = O.rle_encode_arr(sym)
    assert counts.tolist() == ec.tolist() and syms.tolist() == es.tolist()


def test_rle_sentinel_quirks(ctx):
    a, b = ord("a"), ord("b")
    cases = [[a, a, None, a], [a, b, None], [a, a, a, None], [None, a], [None], [None, None],
             [a, None, None, b], [None, None, a, a, None, b, b, b, None, None]]
    rng = np.random.default_rng(5)
    for _ in range(30):
        n = int(rng.integers(1, 9000))
        x = rng.integers(-1, 3, n).tolist()
        cases.append([None if v < 0 else v for v in x])
    for x in cases:
        arr = O.arr_of(x)
        counts, syms = ctx.rle_encode_sym(arr)
        ec, es = O.rle_encode_arr(arr)
        assert counts.tolist() == ec.tolist() and syms.tolist() == es.tolist(), x[:20]


def test_rle_capacity_error(ctx):
    import textcomp
    with pytest.raises(textcomp.TcError) as e:
        ctx.rle_encode_sym(np.array([1, 2, 3, 4], dtype=np.int16), cap=2)
    assert e.value.code == -2


@pytest.mark.parametrize("t", TEXTS, ids=_ids)
def test_fused_encode(ctx, t):
    blk = ctx.encode(t)
    _, eprim, sym = _expect_bwt(t)
    eidx, efl = O.mtf_encode_arr(sym)
    ec, ev = O.rle_encode_u32_arr(eidx)
    assert blk["n"] == len(t) and blk["primary"] == eprim and blk["sigma"] == len(efl)
    assert blk["final_list"].tolist() == efl.tolist()
    assert blk["run_count"].tolist() == ec.tolist() and blk["run_value"].tolist() == ev.tolist()


def test_fused_empty(ctx):
    blk = ctx.encode(b"")
    assert blk["n"] == 0 and len(blk["run_count"]) == 0 and blk["sigma"] == 0


@pytest.mark.parametrize("name,t", [
    ("allA", b"A" * 70000), ("acgt_k", b"ACGT" * 20000),
    ("n_runs", b"N" * 30000 + b"A" + b"N" * 30000 + b"ACGT" * 100),
    ("two_level", (b"AB" * 5000 + b"C") * 5)])
def test_adversarial_repetitive(ctx, name, t):
    sa = ctx.suffix_array(t)
    assert sa.tolist() == O.suffix_array(t).tolist()
    blk = ctx.encode(t)
    _, eprim, sym = _expect_bwt(t)
    eidx, efl = O.mtf_encode_arr(sym)
    ec, ev = O.rle_encode_u32_arr(eidx)
    assert blk["primary"] == eprim
    assert blk["run_count"].tolist() == ec.tolist() and blk["run_value"].tolist() == ev.tolist()


@pytest.mark.parametrize("gen,seed,n", [("ascii", 0xC1, 65536), ("acgtn", 0xC2, 1 << 20),
                                        ("acgtn", 0xC2, 1 << 24)])
def test_configs(ctx, gen, seed, n):
    """BASELINE.json configs[0] (64 KiB ASCII) and configs[1] (16 MiB ACGTN), bit-exact."""
    t = (O.gen_ascii if gen == "ascii" else O.gen_acgtn)(seed, n)
    blk = ctx.encode(t)
    _, eprim, sym = _expect_bwt(t)
    eidx, efl = O.mtf_encode_arr(sym)
    ec, ev = O.rle_encode_u32_arr(eidx)
    assert blk["primary"] == eprim and blk["final_list"].tolist() == efl.tolist()
    assert np.array_equal(blk["run_count"], ec.astype(np.uint32))
    assert np.array_equal(blk["run_value"], ev.astype(np.uint16))


def test_dense_and_sparse_rank_modes_agree(ctx, monkeypatch):
    """The tied-suffix refinement has two rank stores (sparse table / dense ISA)."""
    t = O.gen_acgtn(77, 300000).tobytes() + b"ACGTACGT" * 3000
    exp = O.suffix_array(t).tolist()
    monkeypatch.setenv("TC_SA_FIELDS", "2")          # short round-0 key: many tied suffixes
    assert ctx.suffix_array(t).tolist() == exp
    monkeypatch.setenv("TC_SA_DENSE", "1")
    assert ctx.suffix_array(t).tolist() == exp
    monkeypatch.delenv("TC_SA_FIELDS")
    assert ctx.suffix_array(t).tolist() == exp


def test_finish_and_full_paths_agree(ctx, monkeypatch):
    """Round 0 has a fast path (partial global sort + wave-local finish) and a full path
    (all passes + group kernel); oversize buckets must fall back transparently."""
    t = O.gen_acgtn(3, 200000).tobytes()
    exp = O.suffix_array(t).tolist()
    for passes in ("1", "2", "3"):                # few global passes: bigger buckets, more ties
        monkeypatch.setenv("TC_SA_GLOBAL_PASSES", passes)
        assert ctx.suffix_array(t).tolist() == exp, passes
    monkeypatch.delenv("TC_SA_GLOBAL_PASSES")
    monkeypatch.setenv("TC_SA_FINISH", "0")       # full path only
    assert ctx.suffix_array(t).tolist() == exp
    monkeypatch.delenv("TC_SA_FINISH")
    rep = b"ACGT" * 50000 + t[:1000]              # long repeats: oversize buckets -> fallback
    assert ctx.suffix_array(rep).tolist() == O.suffix_array(rep).tolist()


def _genome_like(seed, n, poly, copies, unit):
    rng = np.random.default_rng(seed)
    t = rng.choice(np.frombuffer(b"ACGT", np.uint8), n)
    fam = rng.choice(np.frombuffer(b"ACGT", np.uint8), unit)
    for p in rng.integers(0, n - unit, copies):
        c = fam.copy()
        mut = rng.random(unit) < 0.1
        c[mut] = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(mut.sum()))
        t[p:p + unit] = c
    for p in rng.integers(0, n - poly, 6):
        t[p:p + rng.integers(poly // 2, poly)] = 65
    p = int(rng.integers(0, n - 400))
    t[p:p + 400] = np.frombuffer(b"CA" * 200, np.uint8)
    return t.tobytes()


@pytest.mark.parametrize("seed,n,poly,copies,unit", [(1, 200000, 500, 40, 120), (2, 70001, 300, 10, 300),
                                                      (3, 300000, 2000, 200, 60), (4, 131072, 90, 5, 1000)])
def test_oversize_buckets_become_tied_groups(ctx, seed, n, poly, copies, unit, monkeypatch):
    """Repeats in otherwise random DNA make a few equal-prefix buckets longer than the finish pass
    ranks in a wave: those buckets are emitted whole as tied groups and ordered by the doubling
    rounds, the rest of the text stays on the fast path (no restart with all passes)."""
    t = _genome_like(seed, n, poly, copies, unit)
    exp = O.suffix_array(t).tolist()
    assert ctx.suffix_array(t).tolist() == exp
    st = ctx.stats()
    assert st.finish_pass == 1 and st.rounds >= 2 and 0 < st.m[1] < (n + 1) // 8
    monkeypatch.setenv("TC_SA_ACCEL_MIN", "0")      # rank lookups through the bitmap / key directory
    assert ctx.suffix_array(t).tolist() == exp
    monkeypatch.setenv("TC_SA_FINISH", "0")         # ... also from the full path (sorted keys)
    assert ctx.suffix_array(t).tolist() == exp
    monkeypatch.delenv("TC_SA_FINISH")
    monkeypatch.delenv("TC_SA_ACCEL_MIN")
    monkeypatch.setenv("TC_SA_TIER2", "0")          # old behaviour: restart on the full path
    assert ctx.suffix_array(t).tolist() == exp
    assert ctx.stats().finish_pass == 0
    monkeypatch.delenv("TC_SA_TIER2")
    blk = ctx.encode(t)
    assert ctx.decode(blk) == t
    _, eprim, sym = _expect_bwt(t)
    eidx, efl = O.mtf_encode_arr(sym)
    ec, ev = O.rle_encode_u32_arr(eidx)
    assert blk["primary"] == eprim
    assert blk["run_count"].tolist() == ec.tolist() and blk["run_value"].tolist() == ev.tolist()


@pytest.mark.parametrize("env", [{"TC_XCD_GROUP": "0"}, {"TC_KEYGEN_FUSED": "0"}, {"TC_KB_ONEHIST": "0"},
                                 {"TC_KEYGEN_FUSED": "0", "TC_KB_ONEHIST": "0"}, {"TC_SA_SAMPLE": "0"},
                                 {"TC_SA_GLOBAL_PASSES": "3"}, {"TC_SA_GLOBAL_PASSES": "5"}, {"TC_SA_FIELDS": "4"},
                                 {"TC_GRID_SCALE_PCT": "7"}, {"TC_RADIX_DIGIT_BITS": "5"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_path_selectors(ctx, env, monkeypatch):
    """Every alternative the host logic can take (single ticket counter = the retry path after a
    look-back timeout, unfused key building, other pass counts / key widths, tiny persistent grids,
    narrower digits) gives the same suffix array and the same encoded block."""
    texts = [O.gen_acgtn(21, 1 << 20).tobytes(), O.gen_ascii(22, 300000).tobytes(),
             _genome_like(5, 150000, 400, 30, 100)]
    ref = [(ctx.suffix_array(t), ctx.encode(t)) for t in texts]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for t, (sa0, blk0) in zip(texts, ref):
        assert np.array_equal(ctx.suffix_array(t), sa0)
        blk = ctx.encode(t)
        assert blk["primary"] == blk0["primary"] and blk["final_list"].tolist() == blk0["final_list"].tolist()
        assert np.array_equal(blk["run_count"], blk0["run_count"]) and np.array_equal(blk["run_value"], blk0["run_value"])


def test_fuzz_small_texts(ctx):
    """300 pseudo-random short texts (alphabets of 1-40 byte values, copied fragments, runs) against the
    oracle: suffix array, fused block and its decode."""
    rng = np.random.default_rng(20261003)
    for it in range(300):
        n = int(rng.integers(1, 2500))
        sigma = int(rng.integers(1, 41))
        alpha = rng.permutation(256)[:sigma]
        t = alpha[rng.integers(0, sigma, n)]
        for _ in range(int(rng.integers(0, 6))):          # repeats: copy a fragment somewhere else
            ln = int(rng.integers(1, max(2, n // 3)))
            a0, b0 = int(rng.integers(0, n - ln + 1)), int(rng.integers(0, n - ln + 1))
            t[b0:b0 + ln] = t[a0:a0 + ln].copy()
        if rng.random() < 0.3:                            # a long run
            ln = int(rng.integers(1, n + 1)); a0 = int(rng.integers(0, n - ln + 1))
            t[a0:a0 + ln] = t[a0]
        tb = t.astype(np.uint8).tobytes()
        assert ctx.suffix_array(tb).tolist() == O.suffix_array(tb).tolist(), (it, n, sigma)
        blk = ctx.encode(tb)
        _, eprim, sym = _expect_bwt(tb)
        eidx, efl = O.mtf_encode_arr(sym)
        ec, ev = O.rle_encode_u32_arr(eidx)
        assert blk["primary"] == eprim and blk["final_list"].tolist() == efl.tolist(), (it, n, sigma)
        assert blk["run_count"].tolist() == ec.tolist() and blk["run_value"].tolist() == ev.tolist(), (it, n, sigma)
        assert ctx.decode(blk) == tb


def _pack_roundtrip(ctx, sigma, counts, vals):
    import ctypes as C
    import torch
    from textcomp import Block
    lib = ctx.lib
    k = len(counts)
    d_c = torch.from_numpy(counts.astype(np.uint32).view(np.int32)).cuda()
    d_v = torch.from_numpy(vals.astype(np.uint16).view(np.int16)).cuda()
    blk = Block(); blk.nruns = k; blk.sigma = sigma
    blk.run_count = d_c.data_ptr(); blk.run_value = d_v.data_ptr()
    bound = lib.tc_block_packed_bound(k, sigma)
    buf = torch.zeros(bound, dtype=torch.uint8, device="cuda")
    nb, ne = C.c_uint64(bound), C.c_uint64()
    assert lib.tc_block_pack_dev(ctx.handle, C.byref(blk), C.c_void_p(buf.data_ptr()), C.byref(nb), C.byref(ne)) == 0
    assert nb.value <= bound
    if k >= 64:
        small = C.c_uint64(k // 4)
        assert lib.tc_block_pack_dev(ctx.handle, C.byref(blk), C.c_void_p(buf.data_ptr()), C.byref(small), C.byref(C.c_uint64())) == -2
        assert small.value > k // 4        # bytes needed
        assert lib.tc_block_pack_dev(ctx.handle, C.byref(blk), C.c_void_p(buf.data_ptr()), C.byref(nb), C.byref(ne)) == 0
    o_c = torch.zeros(k, dtype=torch.int32, device="cuda"); o_v = torch.zeros(k, dtype=torch.int16, device="cuda")
    out = Block(); out.nruns = k; out.run_count = o_c.data_ptr(); out.run_value = o_v.data_ptr()
    assert lib.tc_block_unpack_dev(ctx.handle, C.c_void_p(buf.data_ptr()), nb.value, k, sigma, ne.value, C.byref(out)) == 0
    assert np.array_equal(o_c.cpu().numpy().view(np.uint32), counts.astype(np.uint32))
    assert np.array_equal(o_v.cpu().numpy().view(np.uint16), vals.astype(np.uint16))
    return nb.value, ne.value, buf


def test_block_pack_unpack_roundtrip(ctx):
    """Packed wire format of the runs (used by the multi-GPU gather): exact inverse, incl. count
    escapes, the nibble stream for sigma <= 6 and the two-byte form for sigma > 16."""
    rng = np.random.default_rng(12)
    byte_cases = [(16, rng.integers(1, 4, 50000), rng.integers(0, 16, 50000)),
                  (7, np.array([1, 14, 15, 16, 100000, 2]), np.array([0, 5, 3, 2, 1, 6])),       # escapes
                  (257, rng.integers(1, 300, 20000), rng.integers(0, 257, 20000)),               # 2 bytes / run
                  (16, np.array([2 ** 31]), np.array([15]))]
    for sigma, counts, vals in byte_cases:
        nb, ne, _ = _pack_roundtrip(ctx, sigma, counts, vals)
        assert ne == int((counts >= (15 if sigma <= 16 else 127)).sum())


def test_block_pack_nibble_stream(ctx):
    """sigma <= 6: 4 bits per run of count 1-2, 8 bits for 3-4, escapes beyond; tiles of 16384 runs
    end on 16-byte boundaries."""
    import ctypes as C
    import torch
    from textcomp import Block
    rng = np.random.default_rng(13)
    geo = lambda k: np.minimum(rng.geometric(0.8, k), 9)
    cases = [(6, np.array([1]), np.array([5])),
             (6, np.array([1, 2, 3, 4, 5, 0, 6, 2 ** 31, 1]), np.array([0, 1, 2, 3, 4, 5, 0, 1, 2])),
             (6, geo(31), rng.integers(0, 6, 31)),
             (5, geo(16384), rng.integers(0, 5, 16384)),              # exactly one packer tile
             (6, geo(16385), rng.integers(0, 6, 16385)),
             (6, np.full(40000, 7), rng.integers(0, 6, 40000)),       # every run escapes
             (6, np.full(70000, 3), rng.integers(0, 6, 70000)),       # every run 2 nibbles
             (2, np.ones(100001, dtype=np.int64), rng.integers(0, 2, 100001)),
             (6, geo(1500000), rng.integers(0, 6, 1500000))]
    for sigma, counts, vals in cases:
        nb, ne, buf = _pack_roundtrip(ctx, sigma, counts, vals)
        assert ne == int(((counts < 1) | (counts > 4)).sum())
        nibbles = len(counts) + int((counts > 2).sum()) + int((counts < 1).sum())
        tiles = -(-len(counts) // 16384)
        assert nibbles / 2 <= nb - 4 * ne <= nibbles / 2 + 16 * tiles
    # a body that does not match its header is malformed, never an out-of-bounds write
    counts, vals = geo(50000), rng.integers(0, 6, 50000)
    nb, ne, buf = _pack_roundtrip(ctx, 6, counts, vals)
    lib = ctx.lib
    k = len(counts)
    o_c = torch.zeros(k, dtype=torch.int32, device="cuda"); o_v = torch.zeros(k, dtype=torch.int16, device="cuda")
    out = Block(); out.nruns = k; out.run_count = o_c.data_ptr(); out.run_value = o_v.data_ptr()
    assert lib.tc_block_unpack_dev(ctx.handle, C.c_void_p(buf.data_ptr()), nb, k - 1, 6, ne, C.byref(out)) == -3
    out.nruns = k
    assert lib.tc_block_unpack_dev(ctx.handle, C.c_void_p(buf.data_ptr()), nb - 16, k, 6, ne, C.byref(out)) == -3
    out.nruns = k
    assert lib.tc_block_unpack_dev(ctx.handle, C.c_void_p(buf.data_ptr()), nb - 3, k, 6, ne, C.byref(out)) == -3


def test_mtf_slow_and_fast_incoming_list_paths(ctx, monkeypatch):
    """sigma <= 16: tiles recover their incoming list by a backward scan (fast) or by the
    summary/scan launches (slow; also the fallback when the scan is ambiguous)."""
    rng = np.random.default_rng(31)
    late = b"A" * 40000 + bytes(rng.choice(list(b"ACGT"), 30000).astype(np.uint8)) + b"N" * 5 + b"T" * 20000
    for t in (O.gen_acgtn(9, 100000).tobytes(), late, b"AB" * 30000):
        L = O.bwt_encode_arr(t)
        prim = int(np.nonzero(L < 0)[0][0])
        Lb = np.where(L < 0, 0, L).astype(np.uint8)
        eidx, efl = O.mtf_encode_arr(L)
        for flag in ("1", "0"):
            monkeypatch.setenv("TC_MTF_FASTIN", flag)
            idx, fl = ctx.mtf_encode(Lb, prim)
            assert idx.tolist() == eidx.tolist() and fl.tolist() == efl.tolist(), flag
        # the raw text as an MTF input: symbols that only appear late make the scan ambiguous
        idx, fl = ctx.mtf_encode(np.frombuffer(t, np.uint8), None)
        e2, f2 = O.mtf_encode_arr(np.frombuffer(t, np.uint8).astype(np.int16))
        assert idx.tolist() == e2.tolist() and fl.tolist() == f2.tolist()


def _wordy(seed, n, vocab=300):
    """Zipf-ish words over a small vocabulary: nearly every suffix stays tied after round 0 (the
    sample calls the text hopeless: ranks are written by the first group pass, dense mode)."""
    rs = np.random.default_rng(seed)
    words = [bytes(rs.integers(97, 123, int(rs.integers(2, 9))).astype(np.uint8)) + b" " for _ in range(vocab)]
    p = 1.0 / np.arange(1, vocab + 1) ** 1.1
    ids = rs.choice(vocab, n // 4, p=p / p.sum())
    return b"".join(words[i] for i in ids)[:n]


@pytest.mark.parametrize("n", [1000, 70000, 1 << 20, (1 << 21) + 12345])
def test_dense_ranks_by_regions(ctx, monkeypatch, n):
    """Dense ranks of a large set are stored through (start, rank) pairs partitioned by regions of the
    rank array (tc_sa.hpp, rank_bin_kernel / rank_scatter_kernel) instead of one random store per member:
    same suffix array as the direct stores, in the three places that write ranks (first group pass of a
    hopeless text, the ranks-only pass once dense mode is chosen, every doubling round)."""
    texts = [_wordy(5, n), O.gen_acgtn(9, n // 2).tobytes() + b"ACGTTGCA" * (n // 16)]
    for t in texts:
        exp = O.suffix_array(t).tolist() if n <= (1 << 20) else None
        monkeypatch.setenv("TC_SA_BIN_MIN_LOG2", "40")       # direct stores
        ref = ctx.suffix_array(t)
        if exp is not None:
            assert ref.tolist() == exp
        monkeypatch.setenv("TC_SA_BIN_MIN_LOG2", "0")        # by regions, whatever the size
        assert np.array_equal(ctx.suffix_array(t), ref)
        monkeypatch.setenv("TC_SA_DENSE", "1")               # full path, dense chosen up front
        assert np.array_equal(ctx.suffix_array(t), ref)
        monkeypatch.setenv("TC_SA_FIELDS", "2")
        assert np.array_equal(ctx.suffix_array(t), ref)
        monkeypatch.delenv("TC_SA_DENSE")
        monkeypatch.delenv("TC_SA_FIELDS")


def test_place_workspace_keeps_results(ctx):
    """tc_ctx_place_workspace re-allocates the workspace (several blocks alive at once, the fastest kept):
    the block it leaves in `out` and every later call give the oracle's results."""
    import ctypes as C
    import torch
    from textcomp import Block
    t = O.gen_acgtn(41, 300000)
    eL, eprim, sym = _expect_bwt(t)
    eidx, efl = O.mtf_encode_arr(sym)
    ec, ev = O.rle_encode_u32_arr(eidx)
    d_text = torch.from_numpy(t.copy()).cuda()
    cap = len(t) + 16
    d_cnt = torch.empty(cap, dtype=torch.int32, device="cuda")
    d_val = torch.empty(cap, dtype=torch.int16, device="cuda")
    blk = Block()
    blk.nruns = cap
    blk.run_count = d_cnt.data_ptr()
    blk.run_value = d_val.data_ptr()
    torch.cuda.synchronize()
    ms, chosen = ctx.place_workspace(d_text.data_ptr(), len(t), blk, tries=3)
    assert 1 <= len(ms) <= 3 and 0 <= chosen < len(ms)
    assert int(blk.primary) == eprim and int(blk.nruns) == len(ec)
    assert np.array_equal(d_cnt[:len(ec)].cpu().numpy().astype(np.uint32), ec)
    assert np.array_equal(d_val[:len(ec)].cpu().numpy().astype(np.uint16), ev)
    b2 = ctx.encode(t)                                   # the context goes on working on the kept block
    assert b2["primary"] == eprim and np.array_equal(b2["run_count"], ec)
    assert ctx.suffix_array(t.tobytes()).tolist() == O.suffix_array(t.tobytes()).tolist()


def test_one_kernel_mtf_rle_and_its_fallback(ctx, monkeypatch):
    """sigma <= 8: MTF and RLE (and, for the container of sigma <= 6, the wire format) are one kernel whose tiles
    recover their incoming list by the backward scan; a last column where two codes are met only near the start
    leaves that scan ambiguous far from it, the kernel raises its flag and the encode runs the separate stages.
    Both ways, block and container, against the oracle and against each other (TC_MTF_RLE=0: never the one kernel)."""
    import torch
    rng = np.random.default_rng(77)
    plain = bytes(rng.choice(list(b"ACGTN"), 300000).astype(np.uint8))
    # 'G' and 'T' once each, at the front: their places in the last column are far from most tiles
    rare = b"GT" + bytes(rng.choice(list(b"AC"), 400000).astype(np.uint8))
    runs = b"A" * 70000 + bytes(rng.choice(list(b"AC"), 50000).astype(np.uint8)) + b"C" * 70001
    for t in (plain, rare, runs):
        L = O.bwt_encode_arr(t)
        eidx, efl = O.mtf_encode_arr(L)
        ec, ev = O.rle_encode_u32_arr(eidx)
        blobs = []
        for sel in ("1", "0"):
            monkeypatch.setenv("TC_MTF_RLE", sel)
            blk = ctx.encode(t)
            assert blk["final_list"].tolist() == efl.tolist(), sel
            assert blk["run_count"].tolist() == ec.tolist() and blk["run_value"].tolist() == ev.tolist(), sel
            n = len(t)
            d_t = torch.from_numpy(np.frombuffer(t, np.uint8).copy()).cuda()
            bound = int(ctx.lib.tc_container_bound(n + 2, 257))
            d_o = torch.full((bound + 16,), 0x5A, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            used = ctx.encode_container_dev(d_t.data_ptr(), n, d_o.data_ptr(), bound)
            blobs.append(d_o[:used].cpu().numpy().tobytes())
            assert ctx.decode_container(blobs[-1]) == t, sel
        assert blobs[0] == blobs[1] == ctx.encode_container(t)
