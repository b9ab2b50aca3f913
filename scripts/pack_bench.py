#!/usr/bin/env python3
"""Time tc_block_pack_dev / tc_block_unpack_dev on the encoded block of an ACGTN record."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import torch, textcomp
from textcomp import Block
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
ctx = textcomp.Context(0); lib = ctx.lib
cap = n + 2
d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
d_cnt = torch.empty(cap, dtype=torch.int32, device="cuda"); d_val = torch.empty(cap, dtype=torch.int16, device="cuda")
assert lib.tc_generate_dev(ctx.handle, 0, 0xC3, n, C.c_void_p(d_text.data_ptr())) == 0
blk = Block(); blk.nruns = cap; blk.run_count = d_cnt.data_ptr(); blk.run_value = d_val.data_ptr()
assert lib.tc_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk)) == 0
pcap = n + n // 4 + 4096
buf = torch.empty(pcap, dtype=torch.uint8, device="cuda")
for it in range(3):
    nb, ne = C.c_uint64(pcap), C.c_uint64()
    t0 = time.perf_counter()
    rc = lib.tc_block_pack_dev(ctx.handle, C.byref(blk), C.c_void_p(buf.data_ptr()), C.byref(nb), C.byref(ne))
    dt = time.perf_counter() - t0
    assert rc == 0, lib.tc_last_error(ctx.handle)
    print("pack: %.3f ms  runs %d -> %d bytes (%.3f B/run, %.3f of n), %d escapes" % (dt * 1e3, blk.nruns, nb.value, nb.value / blk.nruns, nb.value / n, ne.value))
o_c = torch.zeros(cap, dtype=torch.int32, device="cuda"); o_v = torch.zeros(cap, dtype=torch.int16, device="cuda")
for it in range(2):
    out = Block(); out.nruns = cap; out.run_count = o_c.data_ptr(); out.run_value = o_v.data_ptr()
    t0 = time.perf_counter()
    rc = lib.tc_block_unpack_dev(ctx.handle, C.c_void_p(buf.data_ptr()), nb.value, blk.nruns, blk.sigma, ne.value, C.byref(out))
    dt = time.perf_counter() - t0
    assert rc == 0, lib.tc_last_error(ctx.handle)
    print("unpack: %.3f ms" % (dt * 1e3))
k = int(blk.nruns)
print("exact:", bool(torch.equal(o_c[:k], d_cnt[:k]) and torch.equal(o_v[:k], d_val[:k])))
