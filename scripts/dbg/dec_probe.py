#!/usr/bin/env python3
"""decode time of one classes_bench class, call by call (is a slow decode the call or its first-time set-up?)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import torch, textcomp
from textcomp import Block
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 30)
ctx = textcomp.Context(0); lib = ctx.lib
g = torch.Generator(device="cuda"); g.manual_seed(7)
t = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device="cuda")[torch.randint(0, 4, (n,), generator=g, device="cuda").long()]
if len(sys.argv) > 1 and sys.argv[1] == "nrun":
    t[n // 3:n // 3 + n // 64] = 78
    for i in range(16):
        a = (2 * i + 1) * (n // 40)
        t[a:a + n // 4096] = 78
t = t.contiguous()
cap = n + 2
d_cnt = torch.empty(cap, dtype=torch.int32, device="cuda"); d_val = torch.empty(cap, dtype=torch.int16, device="cuda")
d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for it in range(4):
    blk = Block(); blk.nruns = cap; blk.run_count = d_cnt.data_ptr(); blk.run_value = d_val.data_ptr()
    t0 = time.perf_counter(); rc = lib.tc_encode_dev(ctx.handle, C.c_void_p(t.data_ptr()), n, C.byref(blk)); t1 = time.perf_counter()
    rc2 = lib.tc_decode_dev(ctx.handle, C.byref(blk), C.c_void_p(d_out.data_ptr())); t2 = time.perf_counter()
    st = ctx.stats()
    print("call %d: encode %.1f ms decode %.1f ms (rc %d %d) ws_grown %d exact %s" % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, rc, rc2, st.ws_grown, bool(torch.equal(d_out, t))), flush=True)
