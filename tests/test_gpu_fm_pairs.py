"""The pair-step FM index (two pattern symbols per lookup, csrc/tc_fm_host.hpp) against the oracle's countFMIndex
(FMIndex/Internal.hs:347-438) on the cases where the two ways of stepping could part: odd and even lengths, bytes
that do not occur in the text at every position of the pattern (Q10: first step => Nothing, later => the loop
stops), ranges that empty between the two symbols of a pair, texts of 1..5 byte values, very short texts; and
locate (FMIndex/Internal.hs:448-542) through the same ranges.  TC_FM_PAIRS=0 (single steps only) must agree too."""
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


def _patterns(rng, text, alpha_all, k):
    pats = []
    n = len(text)
    for _ in range(k):
        r = rng.random()
        m = int(rng.integers(1, 14))
        if r < 0.45 and n >= m:                       # a substring
            o = int(rng.integers(0, n - m + 1))
            p = bytearray(text[o:o + m])
        else:                                         # random letters of the text's alphabet (often absent as a string)
            p = bytearray(rng.choice(np.frombuffer(bytes(sorted(set(text)) or b"A"), np.uint8), m).tobytes())
        if rng.random() < 0.35:                       # a byte that may not occur in the text, anywhere in the pattern
            p[int(rng.integers(0, len(p)))] = int(rng.choice(np.frombuffer(alpha_all, np.uint8)))
        pats.append(bytes(p))
    pats += [b"", text[-1:], text[:1], text[-2:], text[:2], text, text + text[:1]]
    return pats


@pytest.mark.parametrize("seed", range(12))
def test_pair_steps_equal_oracle(ctx, seed, monkeypatch):
    rng = np.random.default_rng(1000 + seed)
    alpha_all = b"ACGTNX#"
    sigma = int(rng.integers(1, 6))
    letters = np.frombuffer(alpha_all[:5], np.uint8)[rng.permutation(5)[:sigma]]
    n = int(rng.choice([1, 2, 3, 5, 17, 449, 5000, 70000]))
    t = letters[rng.integers(0, sigma, n)].astype(np.uint8)
    if n > 100 and rng.random() < 0.5:                # repeats: wide ranges survive many steps
        a0 = int(rng.integers(0, n // 2)); ln = int(rng.integers(1, n // 2))
        t[n - ln:] = t[a0:a0 + ln]
    text = t.tobytes()
    pats = _patterns(rng, text, alpha_all, 300)
    ofm = O.FMIndex(text)
    want = [ofm.count(p) or 0 for p in pats]
    for pairs in ("1", "0"):
        monkeypatch.setenv("TC_FM_PAIRS", pairs)
        fm = ctx.fm_build(text)
        assert fm.count(pats).tolist() == want, "TC_FM_PAIRS=%s" % pairs
        loc = fm.locate(pats[:80])
        for p, hits in zip(pats[:80], loc):
            assert hits.tolist() == ofm.locate(p), p      # 1-based positions in SA order
        fm.close()
