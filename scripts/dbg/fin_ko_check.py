"""debug: the last column by the generic finish kernel against the key-only one (TC_MSD_FINISH_KO), same text"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, textcomp
import oracle as O
os.environ["TC_SA_MSD_MIN_LOG2"] = "10"
ctx = textcomp.Context(0); lib = ctx.lib
def bwt(t):
    n = len(t); d = torch.from_numpy(t).cuda(); L = torch.zeros(n + 1 + 16, dtype=torch.uint8, device="cuda"); p = C.c_uint64()
    torch.cuda.synchronize()
    assert lib.tc_bwt_encode_dev(ctx.handle, C.c_void_p(d.data_ptr()), n, C.c_void_p(L.data_ptr()), C.byref(p)) == 0
    st = ctx.stats()
    return L[:n + 1].cpu().numpy(), p.value, (st.msd_path, st.msd_keyonly)
rng = np.random.default_rng(5)
for n, copies in ((400000, 0), (400000, 30), (1 << 22, 0), (1 << 24, 0)):
    t = np.frombuffer(b"ACGTN", np.uint8)[rng.integers(0, 5, n)].copy()
    for _ in range(copies):
        ln = int(rng.integers(22, 300)); a, b = int(rng.integers(0, n - ln)), int(rng.integers(0, n - ln)); t[b:b + ln] = t[a:a + ln].copy()
    os.environ["TC_MSD_FINISH_KO"] = "0"; L0, p0, s0 = bwt(t)
    os.environ["TC_MSD_FINISH_KO"] = "1"; L1, p1, s1 = bwt(t)
    bad = np.nonzero(L0 != L1)[0]
    print("n %d copies %d: stats old %s new %s primary %d %d, L mismatches %d %s" % (n, copies, s0, s1, p0, p1, len(bad), bad[:8].tolist()), flush=True)
