import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import numpy as np, textcomp
os.environ["TC_SA_MSD_MIN_LOG2"] = "10"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
mode = sys.argv[2] if len(sys.argv) > 2 else "encode"
rng = np.random.default_rng(1)
t = np.frombuffer(b"ACGTN", np.uint8)[rng.integers(0, 5, n)].copy()
ctx = textcomp.Context(0)
print("ctx ok", flush=True)
if mode in ("sa", "both"):
    sa = ctx.suffix_array(t); print("sa ok", ctx.stats().msd_path, flush=True)
if mode in ("encode", "both"):
    blk = ctx.encode(t); print("encode ok", ctx.stats().msd_path, blk["primary"], flush=True)
    assert ctx.decode(blk) == t.tobytes(); print("roundtrip ok", flush=True)
