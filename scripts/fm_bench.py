"""BASELINE configs[3]: FM-index count, 10M x 100-byte ACGTN patterns against the index of a
256 MiB text on one MI355X (99 % substrings of the text, 1 % iid: the miss path).  Prints build
time, count time (patterns resident in HBM), patterns/s and the algorithmic GB/s of SURVEY 8(d):
steps x 2 lookups x 64 B."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, textcomp

n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 28)
npat = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
m = 100
ctx = textcomp.Context(0); lib = ctx.lib
d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
assert lib.tc_generate_dev(ctx.handle, 0, 0xC4, n, C.c_void_p(d_text.data_ptr())) == 0
text = d_text.cpu().numpy()
t0 = time.perf_counter(); fm = ctx.fm_build(text); t_build = time.perf_counter() - t0
rng = np.random.default_rng(0xC4F0)
offs = rng.integers(0, n - m, npat)
idx = torch.from_numpy(offs).cuda()[:, None] + torch.arange(m, device="cuda")[None, :]
pats = d_text[idx.reshape(-1)].reshape(npat, m).contiguous()
miss = torch.arange(99, npat, 100, device="cuda")
d_rand = torch.empty(len(miss) * m, dtype=torch.uint8, device="cuda")
lib.tc_generate_dev(ctx.handle, 0, 0xC4F1, len(miss) * m, C.c_void_p(d_rand.data_ptr()))
pats[miss] = d_rand.reshape(-1, m)
d_offs = (torch.arange(npat + 1, device="cuda", dtype=torch.int64) * m).contiguous()
d_out = torch.zeros(npat, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
lib.tc_fm_count_dev.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p]
best = 1e9
for it in range(4):
    t0 = time.perf_counter()
    rc = lib.tc_fm_count_dev(ctx.handle, fm._h, C.c_void_p(pats.data_ptr()), C.c_void_p(d_offs.data_ptr()), npat, C.c_void_p(d_out.data_ptr()))
    dt = time.perf_counter() - t0
    assert rc == 0
    if it: best = min(best, dt)
out = d_out.cpu().numpy()
hits = int((out > 0).sum())
steps = npat * 0.99 * m + npat * 0.01 * 14
print("n=%d npat=%d build %.1f ms | count %.2f ms -> %.1f Mpatterns/s, %.0f GB/s algorithmic (steps x 2 x 64 B) | found %d (%.2f%%)"
      % (n, npat, t_build * 1e3, best * 1e3, npat / best / 1e6, steps * 128 / best / 1e9, hits, 100.0 * hits / npat))
# spot check against naive counting on a few patterns
tb = text.tobytes()
for j in (0, 1, 99, 12345):
    p = pats[j].cpu().numpy().tobytes()
    c, k = 0, tb.find(p)
    while k >= 0:
        c += 1; k = tb.find(p, k + 1)
    assert c == int(out[j]), (j, c, int(out[j]))
print("spot check vs naive substring count ok")
