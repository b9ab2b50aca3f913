"""Full-size parity pin (CPU only; test infrastructure).  The oracle (oracle/tc_oracle.c, the CPU
restatement of BWT/Internal.hs:110-134, MTF/Internal.hs:128-175 and the RLE of the index stream)
encodes the benchmark record itself -- seed 0xC3, n = 2^30 (BASELINE configs[2]) -- and the 16 MiB
record of configs[1], and writes a DIGEST of the result to tests/golden/c3_digest.json: primary,
sigma, final MTF list, number of runs and position-dependent 64-bit checksums of the last column,
run_count[] and run_value[] (the function of checksum64_kernel in csrc/textcomp.hip, restated in
numpy below).  tests/test_gpu_fullsize.py asserts that the device produces the same digest, which
closes the gap between "round trip exact at 1 GiB" and "bit-exact against the oracle at 1 GiB".

Run once (about 25 GB of memory, several minutes per GiB):  python tests/long/parity_digest.py
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402

GOLD = np.uint64(0x9E3779B97F4A7C15)
M64 = (1 << 64) - 1


def checksum64(a):
    """numpy restatement of checksum64_kernel / checksum64_device: `a` is viewed as little-endian
    u32 words (its byte length must be a multiple of 4)."""
    w = np.ascontiguousarray(a).view(np.uint8).view(np.uint32)
    acc = 0
    step = 1 << 24
    with np.errstate(over="ignore"):
        for lo in range(0, len(w), step):
            x = w[lo:lo + step].astype(np.uint64)
            i = np.arange(lo, lo + len(x), dtype=np.uint64)
            z = ((x << np.uint64(32)) | (i & np.uint64(0xFFFFFFFF))) + (i >> np.uint64(32)) * GOLD
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            acc = (acc + int(z.sum(dtype=np.uint64))) & M64
    return acc ^ ((w.nbytes * 0x9E3779B97F4A7C15) & M64)


def digest(seed, n, log=print):
    d = digest_text(O.gen_acgtn(seed, n), log)
    d["seed"] = seed
    return d


def digest_text(text, log=print):
    """digest of the ORACLE's encode of `text` (a numpy uint8 array)"""
    t0 = time.time()
    n = len(text)
    N = n + 1
    L = np.empty(N, dtype=np.int16)
    assert O.lib().orc_bwt_encode(O._p(text), n, O._p(L)) == N
    log("  n=%d: suffix sort + last column %.0f s" % (n, time.time() - t0))
    primary = int(np.flatnonzero(L < 0)[0])
    Lb = L.astype(np.uint8)          # device form: u8 L[N] with byte 0 in the primary slot
    Lb[primary] = 0
    pad = (-N) % 4
    l_sum = checksum64(np.concatenate([Lb, np.zeros(pad, np.uint8)]))
    del Lb, text
    idx = np.empty(N, dtype=np.int32)
    fl = np.empty(257, dtype=np.int16)
    sigma = O.lib().orc_mtf_encode(O._p(L), N, O._p(idx), O._p(fl))
    del L
    counts = np.empty(N + 1, dtype=np.int64)
    vals = np.empty(N + 1, dtype=np.int32)
    k = int(O.lib().orc_rle_encode_u32(O._p(idx), N, O._p(counts), O._p(vals)))
    del idx
    assert int(counts[:k].sum()) == N and int(counts[:k].max()) < (1 << 32)
    c32 = counts[:k].astype(np.uint32)
    del counts
    v16 = np.zeros(k + (k & 1), dtype=np.uint16)
    v16[:k] = vals[:k]
    del vals
    d = {"n": n, "primary": primary, "sigma": int(sigma), "final_list": [int(v) for v in fl[:sigma]],
         "nruns": k, "last_column_checksum64": "%016x" % l_sum,
         "run_count_checksum64": "%016x" % checksum64(c32), "run_value_checksum64": "%016x" % checksum64(v16),
         "max_run": int(c32.max())}
    log("  n=%d: done in %.0f s: %s" % (n, time.time() - t0, d))
    return d


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [1 << 24, 1 << 30]
    out = os.path.join(ROOT, "tests", "golden", "c3_digest.json")
    res = json.load(open(out)) if os.path.exists(out) else {}
    res["_doc"] = ("digests of the ORACLE's BWT->MTF->RLE encode of gen_acgtn(seed, n); written by "
                   "tests/long/parity_digest.py; checksum64 = checksum64_kernel of csrc/textcomp.hip")
    for n in sizes:
        seed = 0xC3 if n == (1 << 30) else 0xC2
        res["n%d" % n] = digest(seed, n)
        json.dump(res, open(out, "w"), indent=1, sort_keys=True)
