#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point tc_encode (pageable host buffers in,
runs out), beside the device-resident tc_encode_dev."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import ctypes as C
import numpy as np, torch, textcomp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
ctx = textcomp.Context(0)
d = torch.empty(n, dtype=torch.uint8, device="cuda")   # the record of the benchmark, generated on the device
assert ctx.lib.tc_generate_dev(ctx.handle, 0, 0xC3, n, C.c_void_p(d.data_ptr())) == 0
t = d.cpu().numpy()
del d
for it in range(3):
    t0 = time.perf_counter(); blk = ctx.encode(t); dt = time.perf_counter() - t0
    st = ctx.stats()
    print("tc_encode (host buffers) n=%d: %.1f ms = %.2f GB/s (device part %.1f ms); runs out %.2f GB" % (
        n, dt * 1e3, n / dt / 1e9, st.ms_total, len(blk["run_count"]) * 6 / 1e9), flush=True)
# the same entry point with page-locked caller buffers (what a host integration should pass)
import ctypes as C, torch
from textcomp import Block
lib = ctx.lib
cap = n + 2
h_text = torch.from_numpy(np.frombuffer(t, np.uint8).copy()).pin_memory()
h_cnt = torch.empty(cap, dtype=torch.int32).pin_memory(); h_val = torch.empty(cap, dtype=torch.int16).pin_memory()
for it in range(3):
    blk = Block(); blk.nruns = cap; blk.run_count = h_cnt.data_ptr(); blk.run_value = h_val.data_ptr()
    t0 = time.perf_counter()
    rc = lib.tc_encode(ctx.handle, C.c_void_p(h_text.data_ptr()), n, C.byref(blk)); dt = time.perf_counter() - t0
    assert rc == 0
    print("tc_encode (pinned host buffers) n=%d: %.1f ms = %.2f GB/s; %d runs" % (n, dt * 1e3, n / dt / 1e9, blk.nruns), flush=True)
# the container entry point: only the compact form comes back
for it in range(2):
    t0 = time.perf_counter(); blob = ctx.encode_container(t); dt = time.perf_counter() - t0
    print("tc_encode_container (host buffers) n=%d: %.1f ms = %.2f GB/s; container %.3f GB" % (n, dt * 1e3, n / dt / 1e9, len(blob) / 1e9), flush=True)
t0 = time.perf_counter(); back = ctx.decode_container(blob); dt = time.perf_counter() - t0
print("tc_decode_container: %.1f ms = %.2f GB/s; exact %s" % (dt * 1e3, n / dt / 1e9, back == t.tobytes()), flush=True)
