import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import numpy as np, torch, textcomp
from textcomp import Block
ctx = textcomp.Context(0); lib = ctx.lib
rng = np.random.default_rng(13)
geo = lambda k: np.minimum(rng.geometric(0.8, k), 9)
cases = [(6, np.full(40000, 7), rng.integers(0, 6, 40000)),
         (6, np.full(70000, 3), rng.integers(0, 6, 70000)),
         (2, np.ones(100001, dtype=np.int64), rng.integers(0, 2, 100001)),
         (6, geo(1500000), rng.integers(0, 6, 1500000))]
for sigma, counts, vals in cases:
    k = len(counts)
    d_c = torch.from_numpy(counts.astype(np.uint32).view(np.int32)).cuda()
    d_v = torch.from_numpy(vals.astype(np.uint16).view(np.int16)).cuda()
    blk = Block(); blk.nruns = k; blk.sigma = sigma; blk.run_count = d_c.data_ptr(); blk.run_value = d_v.data_ptr()
    bound = lib.tc_block_packed_bound(k, sigma)
    buf = torch.zeros(bound, dtype=torch.uint8, device="cuda")
    nb, ne = C.c_uint64(bound), C.c_uint64()
    rc = lib.tc_block_pack_dev(ctx.handle, C.byref(blk), C.c_void_p(buf.data_ptr()), C.byref(nb), C.byref(ne))
    small = C.c_uint64(k // 4)
    rcs = lib.tc_block_pack_dev(ctx.handle, C.byref(blk), C.c_void_p(buf.data_ptr()), C.byref(small), C.byref(C.c_uint64()))
    nb0 = nb.value
    rc = lib.tc_block_pack_dev(ctx.handle, C.byref(blk), C.c_void_p(buf.data_ptr()), C.byref(nb), C.byref(ne))
    print("small rc", rcs, small.value, "repack rc", rc, nb0, nb.value, lib.tc_last_error(ctx.handle))
    o_c = torch.zeros(k, dtype=torch.int32, device="cuda"); o_v = torch.zeros(k, dtype=torch.int16, device="cuda")
    out = Block(); out.nruns = k; out.run_count = o_c.data_ptr(); out.run_value = o_v.data_ptr()
    rc2 = lib.tc_block_unpack_dev(ctx.handle, C.c_void_p(buf.data_ptr()), nb.value, k, sigma, ne.value, C.byref(out))
    oc = o_c.cpu().numpy().view(np.uint32); ov = o_v.cpu().numpy().view(np.uint16)
    badc = np.nonzero(oc != counts.astype(np.uint32))[0]; badv = np.nonzero(ov != vals.astype(np.uint16))[0]
    print("k", k, "rc", rc, rc2, "nb", nb.value, "ne", ne.value, "badc", len(badc), badc[:8], "badv", len(badv), badv[:8])
    if len(badc) and k < 100:
        print(" counts", counts.tolist()); print(" got   ", oc.tolist()); print(" body", bytes(buf[:nb.value].cpu().numpy()).hex())
    elif len(badc):
        i = badc[0]; print(" around", i, counts[max(0,i-4):i+5].tolist(), oc[max(0,i-4):i+5].tolist())
