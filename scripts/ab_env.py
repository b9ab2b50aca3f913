"""A/B of run-time selectors inside ONE context (same workspace, same step-time mode): alternate the settings step by
step, report the mean 1 GiB encode time per setting.  usage: ab_env.py VAR=a,b [VAR2=c,d ...] (each VAR on its own)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import torch, textcomp
from textcomp import Block
n = 1 << 30
ctx = textcomp.Context(0); lib = ctx.lib
d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
cap = n + 2
d_cnt = torch.empty(cap, dtype=torch.int32, device="cuda"); d_val = torch.empty(cap, dtype=torch.int16, device="cuda")
assert lib.tc_generate_dev(ctx.handle, 0, 0xC3, n, C.c_void_p(d_text.data_ptr())) == 0
torch.cuda.synchronize()
blk = Block()
def step():
    blk.nruns = cap; blk.run_count = d_cnt.data_ptr(); blk.run_value = d_val.data_ptr()
    t0 = time.perf_counter()
    assert lib.tc_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk)) == 0
    return (time.perf_counter() - t0) * 1e3
for _ in range(3): step()
for spec in sys.argv[1:]:
    var, vals = spec.split("=")
    vals = vals.split(",")
    acc = {v: [] for v in vals}
    for rep in range(12):
        for v in vals:
            os.environ[var] = v
            acc[v].append(step())
    os.environ.pop(var, None)
    print(var, " | ".join("%s: mean %.3f min %.3f ms" % (v, sum(a) / len(a), min(a)) for v, a in acc.items()), flush=True)
