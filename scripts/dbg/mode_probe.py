"""The 1 GiB step has two modes, 28.9 and 31.0 ms (partition levels 2 and 3: 4.4 + 5.4 against 5.5 + 6.3 ms),
fixed for the life of a context.  What decides it?
(a) one context, workspace re-allocated (grown, so the same block extended) between measurements: the mode stays;
(b) a new context (new stream, new workspace) each time: the mode changes, often with period 4;
(c) where a one-per-CU grid lands (tc_dbg_dispatch_probe): 32 workgroups per XCC, 256 distinct CUs, XCC =
    blockIdx % 8 in both modes.
Round-2 findings: not the stream (four streams of one context measure alike), not its priority, not the
placement of the workgroups, not the distance between the ping-pong buffers (padding them changes nothing
systematic), not the 2^25-byte spacing of the workgroups' slices (240, 248, 250 workgroups instead of 256:
scripts/dbg/grid_probe.sh), and a plain copy between the same buffers runs at 5.0 TB/s in both modes: it is where the
workspace lands in physical memory, it shows only in the kernels that read 13 GB and write 13 GB at once,
and nothing this library controls selects it."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import torch, textcomp
from textcomp import Block
n = 1 << 30
dev = torch.device("cuda:0")
cap = n + 16
d_text = torch.empty(n, dtype=torch.uint8, device=dev)
d_cnt = torch.empty(cap, dtype=torch.int32, device=dev)
d_val = torch.empty(cap, dtype=torch.int16, device=dev)
ctx = textcomp.Context(0)
lib = ctx.lib
assert lib.tc_generate_dev(ctx.handle, 0, 0xC3, n, C.c_void_p(d_text.data_ptr())) == 0
torch.cuda.synchronize()
f = lib.tc_dbg_stream_bench
f.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
def steps(c):
    blk = Block(); ts = []
    for k in range(3):
        blk.nruns = cap; blk.run_count = d_cnt.data_ptr(); blk.run_value = d_val.data_ptr()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        assert c.lib.tc_encode_dev(c.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk)) == 0
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return "%.2f %.2f" % (ts[1], ts[2])
print("(a) one context, one stream; the workspace grows (is re-allocated) between lines")
for i in range(10):
    print("   workspace #%d: %s ms" % (i, steps(ctx)), flush=True)
    g = C.c_double()
    assert f(ctx.handle, (24 << 30) + (i << 30), 16, 2, 1, C.byref(g)) == 0     # needs 2 x bytes: forces a larger block
ctx.close()
print("(b) a new context (new stream, new workspace) per line")
for i in range(10):
    c = textcomp.Context(0)
    print("   context #%d: %s ms" % (i, steps(c)), flush=True)
    c.close()
print("(c) where a 256-workgroup, one-per-CU grid lands, per context (with its step time)")
import collections, numpy as np
g = lib.tc_dbg_dispatch_probe
g.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
for i in range(10):
    c = textcomp.Context(0)
    t = steps(c)
    out = np.zeros(256 * 6, dtype=np.uint32)
    assert g(c.handle, 256, 147456, 200000, out.ctypes.data) == 0
    o = out.reshape(256, 6)
    xcc = collections.Counter(int(x) for x in o[:, 0])
    cus = collections.Counter((int(r[0]), int(r[1]) & 0xff00 | (int(r[1]) >> 13 & 7) << 16) for r in o)
    start = o[:, 2].astype(np.int64) + (o[:, 3].astype(np.int64) << 32)
    late = int(np.sum(start - start.min() > 500))      # started > 5 us after the first (100 MHz clock)
    order = ["%d:%d.%d.%d" % (int(r[0]), int(r[1]) >> 13 & 7, int(r[1]) >> 12 & 1, int(r[1]) >> 8 & 15) for r in o[:24:8]] + ["|"] + \
            ["%d.%d.%d" % (int(r[1]) >> 13 & 7, int(r[1]) >> 12 & 1, int(r[1]) >> 8 & 15) for r in o[0:256:8][:32]]
    print("   context #%d: %s ms | per XCC %s | distinct CUs %d, most on one CU %d | late starts %d | xcc:se.sh.cu of workgroups 0, 8, 16 | se.sh.cu of XCC 0's workgroups in order: %s" % (
        i, t, [xcc.get(k, 0) for k in range(8)], len(cus), max(cus.values()), late, order), flush=True)
    c.close()
