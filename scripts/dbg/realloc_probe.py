"""Is the fast / slow mode of the partition levels a property of the workspace allocation?  One process, one
input, the context (and with it the workspace) created again and again; step time per context."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import torch, textcomp
from textcomp import Block
n = 1 << 30
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
cap = n + 16
d_text = torch.empty(n, dtype=torch.uint8, device=dev)
d_cnt = torch.empty(cap, dtype=torch.int32, device=dev)
d_val = torch.empty(cap, dtype=torch.int16, device=dev)
ctx0 = textcomp.Context(0)
assert ctx0.lib.tc_generate_dev(ctx0.handle, 0, 0xC3, n, C.c_void_p(d_text.data_ptr())) == 0
torch.cuda.synchronize()
ctx0.close()
hold = []
for r in range(reps):
    ctx = textcomp.Context(0)
    lib = ctx.lib
    blk = Block()
    ts = []
    for k in range(4):
        blk.nruns = cap; blk.run_count = d_cnt.data_ptr(); blk.run_value = d_val.data_ptr()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = lib.tc_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk))
        assert rc == 0
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("context %d: steps %s ms" % (r, " ".join("%.2f" % t for t in ts[1:])), flush=True)
    if os.environ.get("HOLD") == "1" and r % 2 == 0:
        hold.append(torch.empty(256 << 20, dtype=torch.uint8, device=dev))   # shift what the next allocation gets
    ctx.close()
