import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import numpy as np, torch, textcomp
from textcomp.synth import c4_patterns_dev
n, npat, m = 1 << 28, int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 100
ctx = textcomp.Context(0); lib = ctx.lib
d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
assert lib.tc_generate_dev(ctx.handle, 0, 0xC4, n, C.c_void_p(d_text.data_ptr())) == 0
torch.cuda.synchronize()
ref = d_text.clone()
text = d_text.cpu().numpy()
fm = ctx.fm_build(text)
print("text intact after fm_build:", bool(torch.equal(ref, d_text)), flush=True)
pats, d_offs = c4_patterns_dev(ctx, d_text, npat, m)
print("text intact after patterns:", bool(torch.equal(ref, d_text)), flush=True)
miss = torch.arange(99, npat, 100, device="cuda")
d_rand = torch.empty(len(miss) * m, dtype=torch.uint8, device="cuda")
lib.tc_generate_dev(ctx.handle, 0, 0xC4F1, len(miss) * m, C.c_void_p(d_rand.data_ptr())); torch.cuda.synchronize()
print("miss rows are the generated ones:", bool(torch.equal(pats[miss], d_rand.reshape(-1, m))), flush=True)
d_out = torch.zeros(npat, dtype=torch.int64, device="cuda")
lib.tc_fm_count_dev.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p]
for it in range(3):
    assert lib.tc_fm_count_dev(ctx.handle, fm._h, C.c_void_p(pats.data_ptr()), C.c_void_p(d_offs.data_ptr()), npat, C.c_void_p(d_out.data_ptr())) == 0
    out = d_out.cpu().numpy()
    ism = (np.arange(npat) % 100) == 99
    print("call", it, "found", int((out > 0).sum()), "misses found", int((out[ism] > 0).sum()), "hits lost", int((out[~ism] == 0).sum()), flush=True)
