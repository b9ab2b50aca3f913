"""debug: one encode of a small ACGTN text with the key-only finish, trace on"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, textcomp
os.environ["TC_SA_MSD_MIN_LOG2"] = "10"; os.environ["TC_SA_TRACE"] = "2"
ctx = textcomp.Context(0); lib = ctx.lib
rng = np.random.default_rng(5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
t = np.frombuffer(b"ACGTN", np.uint8)[rng.integers(0, 5, n)].copy()
d = torch.from_numpy(t).cuda(); L = torch.zeros(n + 17, dtype=torch.uint8, device="cuda"); p = C.c_uint64()
torch.cuda.synchronize()
assert lib.tc_bwt_encode_dev(ctx.handle, C.c_void_p(d.data_ptr()), n, C.c_void_p(L.data_ptr()), C.byref(p)) == 0
