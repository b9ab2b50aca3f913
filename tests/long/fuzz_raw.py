"""Long randomized run of the stage-level entry points on ARBITRARY symbol streams (any number of
Nothings, any order -- not only BWTs) against the CPU oracle: MTF encode / decode, RLE encode / decode
(quirks Q5-Q8), inverse BWT on arbitrary sequences (Q9).  usage: fuzz_raw.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402
import textcomp  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = textcomp.Context(0)
bad = 0


def outcome(f, exc):
    try:
        return f()
    except exc as e:  # noqa: BLE001
        return "malformed"


for it in range(cases):
    n = int(rng.integers(1, 40000)) if rng.random() < 0.6 else int(rng.integers(1, 200))
    sigma = int(rng.integers(1, 6)) if rng.random() < 0.5 else int(rng.integers(6, 257))
    alpha = rng.permutation(256)[:sigma]
    if rng.random() < 0.5:
        sym = alpha[rng.integers(0, sigma, n)].astype(np.int16)
    else:
        p = 1.0 / np.arange(1, sigma + 1) ** 1.5
        sym = alpha[rng.choice(sigma, n, p=p / p.sum())].astype(np.int16)
    if rng.random() < 0.4:                                        # runs
        sym = np.repeat(sym, rng.integers(1, 9, len(sym)))[:n]
        n = len(sym)
    k = rng.random()
    nn = 0 if k < 0.2 else 1 if k < 0.6 else int(rng.integers(2, 6))  # Nothings
    for _ in range(nn):
        sym[int(rng.integers(0, n))] = -1
    if rng.random() < 0.2:
        sym[-1] = -1
    if rng.random() < 0.2:
        sym[0] = -1
    try:
        eidx, efl = O.mtf_encode_arr(sym)
        idx, fl = ctx.mtf_encode_sym(sym)
        assert np.array_equal(idx, eidx) and fl.tolist() == efl.tolist(), "mtf encode"
        assert np.array_equal(ctx.mtf_decode(eidx, efl), sym), "mtf decode"
        ec, es = O.rle_encode_arr(sym)
        c, s = ctx.rle_encode_sym(sym)
        assert np.array_equal(c, ec) and np.array_equal(s, es), "rle encode"
        assert np.array_equal(ctx.rle_decode(ec.astype(np.uint32), es), O.rle_decode_arr(ec, es)), "rle decode"
        exp = outcome(lambda: O.bwt_decode_arr(sym), O.OracleMalformed)
        got = outcome(lambda: ctx.bwt_decode_sym(sym), textcomp.TcMalformed)
        assert got == exp, "inverse bwt on an arbitrary sequence: %r vs %r" % (got if got == "malformed" else len(got), exp if exp == "malformed" else len(exp))
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("CASE %d n %d sigma %d nothings %d: %s %r" % (it, n, sigma, int((sym < 0).sum()), type(e).__name__, e), flush=True)
        np.save("gpurun_out/fuzzraw_fail_%d_%d.npy" % (seed, it), sym)
        if bad >= 6:
            break
    if it % 200 == 0:
        print("case", it, "failures", bad, flush=True)
print("done: %d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
