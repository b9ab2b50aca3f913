"""Round trips (fused block and chunked stream) at 2^24, 2^26, 2^28 -1/0/+1 bytes, iid ACGTN and random bytes; the oracle
cannot run there: decode(encode) = text and the run lengths sum to N."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O, textcomp
ctx = textcomp.Context(0)
bad = 0
for k in (24, 26, 28):
    for d in (-1, 0, 1):
        n = (1 << k) + d
        for kind in ("acgtn", "bytes"):
            t = O.gen_acgtn(k + d + 50, n) if kind == "acgtn" else np.random.default_rng(k + d).integers(0, 256, n).astype(np.uint8)
            tb = t.tobytes()
            blk = ctx.encode(tb)
            ok = int(blk["run_count"].astype(np.int64).sum()) == n + 1 and ctx.decode(blk) == tb
            blob = ctx.encode_stream(tb, 1 << 23 if k == 24 else 0)
            ok = ok and ctx.decode_stream(blob) == tb
            print(n, kind, "ok" if ok else "FAIL", flush=True)
            bad += 0 if ok else 1
print("failures", bad)
