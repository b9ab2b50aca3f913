#!/usr/bin/env python3
"""Repetitive but not periodic: a Fibonacci word (h + 1 distinct factors of length h: every doubling round keeps nearly all of
the text tied) and a Thue-Morse word -- rounds and time with the chain rounds on (default) and off (TC_SA_CHAIN=0)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import numpy as np, torch, textcomp
from textcomp import Block
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << lg
def fib(n):
    a, b = np.array([65], np.uint8), np.array([65, 67], np.uint8)
    while len(b) < n:
        a, b = b, np.concatenate([b, a])
    return b[:n]
def thue(n):
    t = np.array([65], np.uint8)
    while len(t) < n:
        t = np.concatenate([t, (t ^ 2)])      # 'A' <-> 'C'
    return t[:n]
ctx = textcomp.Context(0); lib = ctx.lib
cap = n + 2
d_cnt = torch.empty(cap, dtype=torch.int32, device="cuda"); d_val = torch.empty(cap, dtype=torch.int16, device="cuda")
d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, mk in (("fibonacci", fib), ("thue_morse", thue)):
    t = torch.from_numpy(mk(n)).cuda()
    for chain in ("1", "0"):
        os.environ["TC_SA_CHAIN"] = chain
        best = 1e9
        for it in range(2):
            blk = Block(); blk.nruns = cap; blk.run_count = d_cnt.data_ptr(); blk.run_value = d_val.data_ptr()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rc = lib.tc_encode_dev(ctx.handle, C.c_void_p(t.data_ptr()), n, C.byref(blk))
            best = min(best, time.perf_counter() - t0)
        st = ctx.stats()
        rc2 = lib.tc_decode_dev(ctx.handle, C.byref(blk), C.c_void_p(d_out.data_ptr()))
        print("%-10s n=2^%d TC_SA_CHAIN=%s: %.1f ms rounds=%d chain_rounds=%d exact=%s m=%s" % (name, lg, chain, best * 1e3, st.rounds, st.chain_rounds,
              bool(torch.equal(d_out, t)), [int(st.m[i]) for i in range(st.rounds)][:10]), flush=True)
