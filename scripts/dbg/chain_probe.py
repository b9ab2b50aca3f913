#!/usr/bin/env python3
"""How many rounds periodic texts of several lengths / periods take, and what each round sheds (chain rounds: tc_chain.hpp)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import torch, textcomp
from textcomp import Block
ctx = textcomp.Context(0); lib = ctx.lib
g = torch.Generator(device="cuda"); g.manual_seed(7)
for lg, pl in [(24, 8), (24, 12), (24, 16), (24, 20), (26, 12), (26, 20), (28, 20), (28, 24)]:
    n = 1 << lg
    base = (torch.randint(0, 4, (1 << pl,), generator=g, device="cuda", dtype=torch.int32) + 65).to(torch.uint8)
    t = base.repeat(n // (1 << pl) + 1)[:n].contiguous()
    cap = n + 2
    d_cnt = torch.empty(cap, dtype=torch.int32, device="cuda"); d_val = torch.empty(cap, dtype=torch.int16, device="cuda")
    blk = Block(); blk.nruns = cap; blk.run_count = d_cnt.data_ptr(); blk.run_value = d_val.data_ptr()
    torch.cuda.synchronize()
    rc = lib.tc_encode_dev(ctx.handle, C.c_void_p(t.data_ptr()), n, C.byref(blk))
    st = ctx.stats()
    print("n=2^%d period=2^%d rc=%d rounds=%d chain=%d dups=%d h=%s shed=%s" % (lg, pl, rc, st.rounds, st.chain_rounds, st.sample_dups,
          [int(st.h[i]) for i in range(st.rounds)][:8], [int(st.m[i]) - int(st.m[i + 1]) for i in range(st.rounds - 1)][:8] + [int(st.m[st.rounds - 1])]), flush=True)
