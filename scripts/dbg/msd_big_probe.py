import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, textcomp, oracle as O
os.environ["TC_SA_MSD_MIN_LOG2"] = "10"; os.environ["TC_SA_MSD_BIG"] = "1"
rng = np.random.default_rng(99); n = 600000
t = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
t[1000:13000] = ord("A")
ctx = textcomp.Context(0)
sa = ctx.suffix_array(t); st = ctx.stats()
print("msd", st.msd_path, "finish", st.finish_pass, "rounds", st.rounds, "m", [int(st.m[i]) for i in range(st.rounds)], "h", [int(st.h[i]) for i in range(st.rounds)])
want = O.suffix_array(t)
bad = np.nonzero(sa.astype(np.int64) != want.astype(np.int64))[0]
print("mismatches", len(bad), bad[:10], sa[bad[:5]], want[bad[:5]])
