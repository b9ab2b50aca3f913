"""Round 0 of the suffix sort as MSD partition levels + bucket finish (csrc/tc_msd.hpp): the path the
1 GiB ACGTN benchmark record takes, forced here at sizes the oracle sorts in seconds
(TC_SA_MSD_MIN_LOG2 lowers the length from which it is chosen).  Whatever the path, the result is
the reference's order of the n+1 suffixes (createSuffixArray, BWT/Internal.hs:110-134) -- bit-exact
against the oracle; `tc_stats.msd_path` tells which way round 0 went."""
import os

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


def _check(ctx, t, expect_msd):
    t = np.ascontiguousarray(t, dtype=np.uint8)
    sa = ctx.suffix_array(t)
    st = ctx.stats()
    assert st.msd_path == (1 if expect_msd else 0), (st.msd_path, st.finish_pass, st.rounds)
    want = O.suffix_array(t)
    assert np.array_equal(sa.astype(np.int64), want.astype(np.int64))
    blk = ctx.encode(t)
    L = O.bwt_encode_arr(t)
    idx, fl = O.mtf_encode_arr(L)
    counts, vals = O.rle_encode_u32_arr(idx)
    assert blk["primary"] == int(np.nonzero(L < 0)[0][0])
    assert blk["final_list"].tolist() == fl.tolist()
    assert np.array_equal(blk["run_count"], counts) and np.array_equal(blk["run_value"], vals)
    assert ctx.decode(blk) == t.tobytes()


@pytest.mark.parametrize("n", [32769, 50000, 100003, (1 << 20) - 1, (1 << 22) + 5])
def test_msd_iid_acgtn(ctx, n, monkeypatch):
    monkeypatch.setenv("TC_SA_MSD_MIN_LOG2", "10")
    _check(ctx, O.gen_acgtn(0xC2 + n, n), True)


@pytest.mark.parametrize("sigma", [2, 3, 4, 7, 15])
def test_msd_other_small_alphabets(ctx, sigma, monkeypatch):
    """fields of 8 / 5 / 3 / 2 / 2 symbols (base sigma + 1 in 8 bits)"""
    monkeypatch.setenv("TC_SA_MSD_MIN_LOG2", "10")
    rng = np.random.default_rng(sigma)
    alpha = rng.permutation(256)[:sigma]
    _check(ctx, alpha[rng.integers(0, sigma, 300000)], True)


def test_msd_with_ties_beyond_the_key(ctx, monkeypatch):
    """copied stretches longer than the 21 key symbols: members equal on all key bits leave the
    bucket finish as tied groups and are ordered by the doubling rounds"""
    monkeypatch.setenv("TC_SA_MSD_MIN_LOG2", "10")
    rng = np.random.default_rng(5)
    t = O.gen_acgtn(77, 400000).copy()
    for _ in range(40):
        ln = int(rng.integers(22, 300))
        a, b = int(rng.integers(0, len(t) - ln)), int(rng.integers(0, len(t) - ln))
        t[b:b + ln] = t[a:a + ln].copy()
    _check(ctx, t, True)
    assert ctx.stats().rounds >= 2


def test_msd_gives_way_to_lsd_on_long_buckets(ctx, monkeypatch):
    """a poly-A tract of 20000 symbols: one level-3 bucket beyond MSDF_CAP -> the LSD way, same result;
    a tract of 1500 stays on the MSD way (its bucket fits a chunk, its members leave as one tied group)"""
    monkeypatch.setenv("TC_SA_MSD_MIN_LOG2", "10")
    t = O.gen_acgtn(78, 300000).copy()
    t[1000:21000] = ord("A")
    _check(ctx, t, False)
    t = O.gen_acgtn(79, 300000).copy()
    t[1000:2500] = ord("A")
    _check(ctx, t, True)


def test_msd_16m_equals_digest_record(ctx, monkeypatch):
    """BASELINE configs[1] (16 MiB ACGTN) through the MSD way"""
    monkeypatch.setenv("TC_SA_MSD_MIN_LOG2", "10")
    _check(ctx, O.gen_acgtn(0xC2, 1 << 24), True)


def test_default_threshold_keeps_small_records_on_lsd(ctx):
    t = O.gen_acgtn(3, 1 << 20)
    ctx.suffix_array(t)
    assert ctx.stats().msd_path == 0


@pytest.mark.parametrize("n,sigma", [(40000, 4), (300000, 4), ((1 << 22) + 3, 4), (300000, 5), (300000, 2)])
def test_msd_big_finish_instance(ctx, n, sigma, monkeypatch):
    """the finish instance for buckets of a few thousand pairs (4-letter DNA at 1 GiB: ~4096 per 9-mer), which
    also writes the keys in final order so that rank lookups are binary searches; forced here at small sizes"""
    monkeypatch.setenv("TC_SA_MSD_MIN_LOG2", "10")
    monkeypatch.setenv("TC_SA_MSD_BIG", "1")
    rng = np.random.default_rng(n + sigma)
    alpha = np.frombuffer(b"ACGTN", np.uint8)[:sigma] if sigma <= 5 else rng.permutation(256)[:sigma]
    t = alpha[rng.integers(0, sigma, n)].copy()
    for _ in range(20):      # copies beyond the 21 key symbols: tied groups, refined through the sorted keys
        ln = int(rng.integers(22, 200))
        a, b = int(rng.integers(0, n - ln)), int(rng.integers(0, n - ln))
        t[b:b + ln] = t[a:a + ln].copy()
    _check(ctx, t, True)


def test_msd_big_instance_keeps_long_buckets(ctx, monkeypatch):
    """a poly-A tract of 12000 symbols and a repeat family: level-3 buckets far above the finish chunk leave the
    big instance as whole tied groups (in place, ordered by the doubling rounds from 9 symbols on) -- the MSD
    way is kept"""
    monkeypatch.setenv("TC_SA_MSD_MIN_LOG2", "10")
    monkeypatch.setenv("TC_SA_MSD_BIG", "1")
    rng = np.random.default_rng(99)
    n = 600000
    t = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
    t[1000:13000] = ord("A")
    fam = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 300)]
    for _ in range(80):
        a = int(rng.integers(30000, n - 400))
        c = fam.copy()
        mut = rng.random(300) < 0.1
        c[mut] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(mut.sum()))]
        t[a:a + 300] = c
    _check(ctx, t, True)
    assert ctx.stats().rounds >= 3


def test_msd_keyonly_levels_and_their_tied_set(ctx, monkeypatch):
    """round 3: an encode asks for no suffix array, so the MSD levels move keys only and the (few) suffixes that stay
    tied beyond the key get their suffix starts back from one pass over the text (tied_probe_kernel).  Copies of
    22 .. 300 symbols make a few thousand ties in groups of two and more; texts of 3 and 16 symbol codes take
    other field widths (5 and 2 symbols per field) through the rolling value."""
    monkeypatch.setenv("TC_SA_MSD_MIN_LOG2", "10")
    rng = np.random.default_rng(17)
    for sigma, n in ((5, 400000), (2, 150000), (15, 200000)):
        alpha = np.sort(rng.permutation(256)[:sigma]).astype(np.uint8)
        t = alpha[rng.integers(0, sigma, n)]
        for _ in range(30):
            ln = int(rng.integers(22, 300))
            a, b = int(rng.integers(0, n - ln)), int(rng.integers(0, n - ln))
            t[b:b + ln] = t[a:a + ln].copy()
        t[n - 40:] = t[1000:1040]          # a tie that reaches the end of the text
        _check(ctx, t, True)               # (suffix array: with suffix starts; encode: keys only)
        st = ctx.stats()                   # (of the decode's... no: of the last ENCODE, ctx.encode in _check)
        blk = ctx.encode(t)
        st = ctx.stats()
        assert st.msd_path == 1 and st.msd_keyonly == 1 and st.rounds >= 2 and st.m[1] > 0
        monkeypatch.setenv("TC_SA_MSD_KEYONLY", "0")
        ref = ctx.encode(t)
        assert ctx.stats().msd_keyonly == 0
        monkeypatch.delenv("TC_SA_MSD_KEYONLY")
        assert blk["primary"] == ref["primary"] and np.array_equal(blk["run_count"], ref["run_count"])
        assert np.array_equal(blk["run_value"], ref["run_value"])


def test_msd_keyonly_gives_way_when_too_many_suffixes_are_tied(ctx, monkeypatch):
    """more ties than the table of the key-only levels is made for (2^15): the levels run once more with the suffix
    starts moving along -- same block"""
    monkeypatch.setenv("TC_SA_MSD_MIN_LOG2", "10")
    monkeypatch.setenv("TC_SA_MSD", "2")       # (the collision sample would send this text to the LSD way)
    rng = np.random.default_rng(23)
    n = 600000
    t = O.gen_acgtn(5, n).copy()
    for _ in range(20):                        # 20 copies of 1000 symbols: ~40 000 tied suffixes (source and copy)
        ln = 1000
        a, b = int(rng.integers(0, n - ln)), int(rng.integers(0, n - ln))
        t[b:b + ln] = t[a:a + ln].copy()
    _check(ctx, t, True)
    ctx.encode(t)
    st = ctx.stats()
    assert st.msd_path == 1 and st.msd_keyonly == 0 and (1 << 15) < st.m[1] < 75000


def test_msd_keyonly_finish_with_a_digit_above_half_the_text(ctx, monkeypatch):
    """a binary text, 60 % of it one run: one field value holds more than half of the level-1 digit counts, so the
    equal-mass bin map of the key-only finish kernel multiplies a share above 1/2 by cumulative shares above 1/2 --
    taken as a signed product (HIP's __umul24 returns int) the bin left the table and the kernel never came back.
    The input is the soak case that showed it (tests/long/fuzz_long.py 80 102 200000, case 50).  (The run also makes
    level-3 buckets above the finish chunk, so the attempt ends with the LSD way: msd_path 0 -- after the kernel ran.)"""
    monkeypatch.setenv("TC_SA_MSD", "2")
    monkeypatch.setenv("TC_SA_MSD_MIN_LOG2", "10")
    t = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "binary_long_run_128k.npz"))["text"]
    _check(ctx, t, False)
