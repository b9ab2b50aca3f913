"""Record lengths at and around the tile sizes of every kernel (2^k - 1, 2^k, 2^k + 1 and N = n + 1 there):
encode against the oracle, decode, container.  usage: boundary_sweep.py [max_log2]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402
import textcomp  # noqa: E402

hi = int(sys.argv[1]) if len(sys.argv) > 1 else 21
ctx = textcomp.Context(0)
rng = np.random.default_rng(99)
bad = 0
for k in range(6, hi + 1):
    for d in (-2, -1, 0, 1):
        n = (1 << k) + d
        for kind in ("acgtn", "ascii", "bytes", "runs"):
            if kind == "acgtn":
                t = O.gen_acgtn(k * 7 + d + 3, n).tobytes()
            elif kind == "ascii":
                t = O.gen_ascii(k * 5 + d + 3, n).tobytes()
            elif kind == "bytes":
                a = rng.integers(0, 256, n).astype(np.uint8)
                if n >= 256:
                    a[rng.permutation(n)[:256]] = np.arange(256)
                t = a.tobytes()
            else:
                t = np.repeat(rng.choice(list(b"ACGT"), n // 8 + 1), 8)[:n].astype(np.uint8).tobytes()
            try:
                blk = ctx.encode(t)
                L = O.bwt_encode_arr(t)
                eidx, efl = O.mtf_encode_arr(L)
                ec, ev = O.rle_encode_u32_arr(eidx)
                assert blk["primary"] == int(np.nonzero(L < 0)[0][0]) and blk["final_list"].tolist() == efl.tolist()
                assert blk["run_count"].tolist() == ec.tolist() and blk["run_value"].tolist() == ev.tolist()
                assert ctx.decode(blk) == t
                assert ctx.decode_container(ctx.encode_container(t)) == t
            except Exception as e:  # noqa: BLE001
                bad += 1
                print("FAIL n=%d kind=%s: %r" % (n, kind, e), flush=True)
    print("2^%d done, failures %d" % (k, bad), flush=True)
print("done, failures", bad)
sys.exit(1 if bad else 0)
