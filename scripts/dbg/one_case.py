"""debug: encode one saved text (npy) with the selectors of the environment, trace on"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, textcomp
t = np.load(sys.argv[1])
print("n", len(t), "sigma", len(np.unique(t)), flush=True)
ctx = textcomp.Context(0)
what = sys.argv[2] if len(sys.argv) > 2 else "encode"
if what == "sa":
    sa = ctx.suffix_array(t.tobytes()); print("sa ok", sa[:4], flush=True)
else:
    blk = ctx.encode(t.tobytes()); print("encode ok", blk["primary"], len(blk["run_count"]), flush=True)
