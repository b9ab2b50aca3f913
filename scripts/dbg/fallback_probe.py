"""debug: does the one-kernel MTF+RLE raise its flag on a last column with codes met only near the start?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, textcomp
rng = np.random.default_rng(77)
ctx = textcomp.Context(0)
for name, t in (("plain", bytes(rng.choice(list(b"ACGTN"), 300000).astype(np.uint8))),
                ("rare", b"GT" + bytes(rng.choice(list(b"AC"), 400000).astype(np.uint8))),
                ("rare2", bytes(rng.choice(list(b"AC"), 200000).astype(np.uint8)) + b"G" + bytes(rng.choice(list(b"AC"), 200000).astype(np.uint8)) + b"T" + bytes(rng.choice(list(b"AC"), 100000).astype(np.uint8)))):
    ctx.encode(t); st = ctx.stats()
    print(name, "ms_mtf %.3f ms_rle %.3f" % (st.ms_mtf, st.ms_rle), flush=True)
