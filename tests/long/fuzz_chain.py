"""Differential run for the CHAIN rounds of the prefix doubling (tc_chain.hpp) against the CPU oracle: texts made of
periods -- a random block repeated, with mutations, a foreign head / tail, several periodic stretches of different
periods, runs, Fibonacci / Thue-Morse words -- so that many suffixes keep seeing the same rank at + h, + 2 h, ...
Run with TC_SA_CHAIN=2 TC_SA_DENSE=1 TC_SA_SEG_MIN=1 (a chain round in every dense round with h >= 4) or with the
defaults (the host's own trigger).  usage: fuzz_chain.py [cases] [seed] [max_n]"""
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
max_n = int(sys.argv[3]) if len(sys.argv) > 3 else 60000
rng = np.random.default_rng(seed)
ctx = textcomp.Context(0)
bad = 0
chained = 0


def fib_word(n):
    a, b = np.array([0], np.uint8), np.array([0, 1], np.uint8)
    while len(b) < n:
        a, b = b, np.concatenate([b, a])
    return b[:n]


def make(n):
    kind = int(rng.integers(0, 7))
    sigma = int(rng.integers(1, 6))
    alpha = rng.permutation(256)[:max(sigma, 2)].astype(np.uint8)
    if kind == 0:      # one block repeated
        per = int(rng.integers(1, max(2, n // 3)))
        t = np.resize(alpha[rng.integers(0, sigma, per)], n)
    elif kind == 1:    # repeated block + point mutations
        per = int(rng.integers(1, max(2, n // 8)))
        t = np.resize(alpha[rng.integers(0, sigma, per)], n).copy()
        k = int(rng.integers(0, 6))
        t[rng.integers(0, n, k)] = alpha[rng.integers(0, len(alpha), k)]
    elif kind == 2:    # foreign head and tail around a periodic middle
        per = int(rng.integers(1, max(2, n // 6)))
        t = np.resize(alpha[rng.integers(0, sigma, per)], n).copy()
        a0 = int(rng.integers(0, n // 4 + 1)); b0 = int(rng.integers(0, n // 4 + 1))
        t[:a0] = alpha[rng.integers(0, len(alpha), a0)]
        if b0: t[n - b0:] = alpha[rng.integers(0, len(alpha), b0)]
    elif kind == 3:    # several periodic stretches (different periods, the same alphabet)
        t = np.empty(n, np.uint8)
        at = 0
        while at < n:
            ln = int(rng.integers(1, max(2, n // 2)))
            per = int(rng.integers(1, max(2, ln // 3 + 1)))
            seg = np.resize(alpha[rng.integers(0, sigma, per)], ln)
            t[at:at + ln] = seg[:n - at]
            at += ln
    elif kind == 4:    # runs
        t = np.repeat(alpha[rng.integers(0, len(alpha), n)], rng.integers(1, 40, n))[:n]
        t = np.resize(t, n)
    elif kind == 5:    # Fibonacci word (every prefix doubling's favourite) over two letters
        t = alpha[:2][fib_word(n)]
    else:              # a period of a period: (u^a v)^b
        u = alpha[rng.integers(0, sigma, int(rng.integers(1, 30)))]
        v = alpha[rng.integers(0, len(alpha), int(rng.integers(0, 30)))]
        blk = np.concatenate([np.tile(u, int(rng.integers(1, 50))), v])
        t = np.resize(blk, n)
    return np.ascontiguousarray(t, dtype=np.uint8)


for it in range(cases):
    n = int(rng.integers(8, max_n)) if rng.random() < 0.7 else int(rng.integers(1, 400))
    t = make(n)
    tb = t.tobytes()
    try:
        sa = ctx.suffix_array(tb)
        chained += 1 if ctx.stats().chain_rounds else 0
        assert sa.tolist() == O.suffix_array(tb).tolist(), "suffix array"
        blk = ctx.encode(tb)
        L = O.bwt_encode_arr(tb)
        eidx, efl = O.mtf_encode_arr(L)
        ec, ev = O.rle_encode_u32_arr(eidx)
        assert blk["primary"] == int(np.nonzero(L < 0)[0][0]) and blk["final_list"].tolist() == efl.tolist(), "header"
        assert blk["run_count"].tolist() == ec.tolist() and blk["run_value"].tolist() == ev.tolist(), "runs"
        assert ctx.decode(blk) == tb, "decode"
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("CASE %d seed %d n %d: %s %r" % (it, seed, n, type(e).__name__, e), flush=True)
        os.makedirs("gpurun_out", exist_ok=True)
        np.save("gpurun_out/fuzz_chain_fail_%d_%d.npy" % (seed, it), t)
        if bad >= 5:
            break
    if it % 100 == 0:
        print("case", it, "ok so far, failures", bad, flush=True)
print("done: %d cases, %d failures, %d with chain rounds" % (cases, bad, chained))
sys.exit(1 if bad else 0)
