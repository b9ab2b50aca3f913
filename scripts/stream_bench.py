"""Chunked stream (tc_encode_stream / tc_decode_stream): host text -> containers -> host text, with
the copies of the neighbouring records overlapped with the device work, against the same records
sent one at a time through tc_encode_container.  usage: stream_bench.py [total_bytes] [block_bytes]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import textcomp  # noqa: E402
import torch  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 4 << 30
block = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 30
ctx = textcomp.Context(0)
lib = ctx.lib
text = np.empty(total, np.uint8)
d = torch.empty(min(block, total), dtype=torch.uint8, device="cuda")
for k in range(0, total, block):                      # generated on the device, record by record
    m = min(block, total - k)
    assert lib.tc_generate_dev(ctx.handle, 0, 0xC500 + k // block, m, C.c_void_p(d.data_ptr())) == 0
    text[k:k + m] = d[:m].cpu().numpy()
del d
cap = 2 * total + (1 << 20)
out = np.empty(cap, np.uint8)
P = lambda a: C.c_void_p(a.ctypes.data)
for rep in range(2):
    used = C.c_uint64(cap)
    t0 = time.perf_counter()
    rc = lib.tc_encode_stream(ctx.handle, P(text), total, block, P(out), C.byref(used))
    t1 = time.perf_counter()
    assert rc == 0, lib.tc_last_error(ctx.handle)
    print("encode_stream   %d x %d MiB: %.1f ms = %.2f GB/s host to host, %.3f bytes out per byte in"
          % (-(-total // block), block >> 20, (t1 - t0) * 1e3, total / (t1 - t0) / 1e9, used.value / total), flush=True)
one = np.empty(2 * block + (1 << 20), np.uint8)
t0 = time.perf_counter()
for k in range(0, total, block):
    m = min(block, total - k)
    u1 = C.c_uint64(len(one))
    assert lib.tc_encode_container(ctx.handle, P(text[k:k + m]), m, P(one), C.byref(u1)) == 0
t1 = time.perf_counter()
print("encode_container, one record at a time: %.1f ms = %.2f GB/s" % ((t1 - t0) * 1e3, total / (t1 - t0) / 1e9), flush=True)
back = np.empty(total, np.uint8)
for rep in range(2):
    got = C.c_uint64(total)
    t0 = time.perf_counter()
    rc = lib.tc_decode_stream(ctx.handle, P(out), used.value, P(back), C.byref(got))
    t1 = time.perf_counter()
    assert rc == 0 and got.value == total, lib.tc_last_error(ctx.handle)
    print("decode_stream: %.1f ms = %.2f GB/s host to host" % ((t1 - t0) * 1e3, total / (t1 - t0) / 1e9), flush=True)
assert np.array_equal(back, text)
print("round trip exact")
