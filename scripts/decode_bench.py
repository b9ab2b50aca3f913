"""Round trip at scale: tc_encode_dev -> tc_decode_dev on a device-resident record; verifies the
decoded bytes equal the input (bit-exact) and times both directions."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import torch, textcomp
from textcomp import Block
n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 30)
ctx = textcomp.Context(0); lib = ctx.lib
d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
lib.tc_generate_dev(ctx.handle, 0, 0xC3, n, C.c_void_p(d_text.data_ptr()))
cap = n + 2
d_cnt = torch.empty(cap, dtype=torch.int32, device="cuda"); d_val = torch.empty(cap, dtype=torch.int16, device="cuda")
blk = Block(); blk.nruns = cap; blk.run_count = d_cnt.data_ptr(); blk.run_value = d_val.data_ptr()
t0 = time.perf_counter(); rc = lib.tc_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk)); te = time.perf_counter() - t0
assert rc == 0, lib.tc_last_error(ctx.handle)
d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
for it in range(2):
    t0 = time.perf_counter(); rc = lib.tc_decode_dev(ctx.handle, C.byref(blk), C.c_void_p(d_out.data_ptr())); td = time.perf_counter() - t0
    assert rc == 0, lib.tc_last_error(ctx.handle)
ok = bool(torch.equal(d_out, d_text))
print("n=%d runs=%d encode(first call) %.1f ms | decode %.1f ms = %.2f GB/s | round trip %s" % (n, blk.nruns, te * 1e3, td * 1e3, n / td / 1e9, "EXACT" if ok else "MISMATCH"))
assert ok
