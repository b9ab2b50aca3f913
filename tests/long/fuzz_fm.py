"""Long randomized run of the FM-index (count / locate) against the CPU oracle and naive matching.
usage: fuzz_fm.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402
import textcomp  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = textcomp.Context(0)
bad = 0
for it in range(cases):
    n = int(rng.integers(1, 30000))
    sigma = int(rng.integers(1, 8)) if rng.random() < 0.6 else int(rng.integers(8, 257))
    alpha = rng.permutation(256)[:sigma]
    t = alpha[rng.integers(0, sigma, n)].astype(np.uint8)
    for _ in range(int(rng.integers(0, 5))):
        ln = int(rng.integers(1, max(2, n // 2)))
        a0, b0 = int(rng.integers(0, n - ln + 1)), int(rng.integers(0, n - ln + 1))
        t[b0:b0 + ln] = t[a0:a0 + ln].copy()
    tb = t.tobytes()
    pats = []
    for _ in range(int(rng.integers(1, 60))):
        k = rng.random()
        m = int(rng.integers(1, 40))
        if k < 0.6:                                     # a substring (count >= 1)
            a0 = int(rng.integers(0, n)); pats.append(tb[a0:a0 + m])
        elif k < 0.8:                                   # random over the alphabet
            pats.append(alpha[rng.integers(0, sigma, m)].astype(np.uint8).tobytes())
        elif k < 0.9:                                   # with a byte that may be absent (Q10)
            p = bytearray(tb[:m] if m <= n else tb); p[int(rng.integers(0, len(p)))] = int(rng.integers(0, 256)); pats.append(bytes(p))
        else:
            pats.append(b"")
    try:
        ofm = O.FMIndex(tb)
        fm = ctx.fm_build(tb)
        counts = fm.count(pats)
        hits = fm.locate(pats)
        fm.close()
        for p, c, h in zip(pats, counts, hits):
            exp = ofm.count(p)
            assert (None if c == 0 else int(c)) == exp, ("count", p, int(c), exp)
            assert sorted(int(v) for v in h) == sorted(ofm.locate(p)), ("locate", p)
            if p and all(bytes([b]) in tb for b in p):  # every byte occurs: equals naive matching
                naive = sum(1 for i in range(n - len(p) + 1) if tb.startswith(p, i))
                assert int(c) == naive, ("naive", p, int(c), naive)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("CASE %d n %d sigma %d: %r" % (it, n, sigma, e), flush=True)
        if bad >= 5:
            break
    if it % 50 == 0:
        print("case", it, "failures", bad, flush=True)
print("done: %d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
