"""BASELINE configs[3]: FM-index count, 10M x 100-byte ACGTN patterns against the index of a
256 MiB text on one MI355X (99 % substrings of the text, 1 % iid: the miss path; generator of
SURVEY 8d in textcomp/synth.py).  Prints build time, count time (patterns resident in HBM),
patterns/s and the algorithmic GB/s of SURVEY 8(d): executed steps x 2 lookups x 64 B + pattern
bytes.  `steps` = the steps each pattern really executes (a miss pattern stops when its range
empties), counted on the host from the result of a per-prefix count, not estimated."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import torch, textcomp
from textcomp.synth import c4_patterns_dev

n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 28)
npat = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
m = 100
ctx = textcomp.Context(0); lib = ctx.lib
d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
assert lib.tc_generate_dev(ctx.handle, 0, 0xC4, n, C.c_void_p(d_text.data_ptr())) == 0
torch.cuda.synchronize()
text = d_text.cpu().numpy()
t0 = time.perf_counter(); fm = ctx.fm_build(text); t_build = time.perf_counter() - t0
pats, d_offs = c4_patterns_dev(ctx, d_text, npat, m)
d_out = torch.zeros(npat, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
lib.tc_fm_count_dev.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p]
ts = []
for it in range(iters + 1):
    t0 = time.perf_counter()
    rc = lib.tc_fm_count_dev(ctx.handle, fm._h, C.c_void_p(pats.data_ptr()), C.c_void_p(d_offs.data_ptr()), npat, C.c_void_p(d_out.data_ptr()))
    dt = time.perf_counter() - t0
    assert rc == 0
    if it: ts.append(dt)
best, mean = min(ts), sum(ts) / len(ts)
out = d_out.cpu().numpy()
hits = int((out > 0).sum())
# executed steps: a hit pattern runs all m steps; a miss runs until its range is empty -- measured by
# counting the miss patterns' suffixes of growing length (the first length with count 0 is the last step)
miss_idx = np.arange(99, npat, 100)
miss_idx = miss_idx[out[miss_idx] == 0]
steps_hit = int((out > 0).sum()) * m
sub = miss_idx[:2000]
steps_miss_mean = 0.0
if len(sub):
    mp = pats[torch.from_numpy(sub).cuda()]
    alive = np.ones(len(sub), bool); last = np.zeros(len(sub), np.int64)
    for L in range(1, 40):
        sfx = mp[:, m - L:].contiguous()
        so = (torch.arange(len(sub) + 1, device="cuda", dtype=torch.int64) * L).contiguous()
        o = torch.zeros(len(sub), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()   # the library runs on its own stream: torch's kernels above must be done
        assert lib.tc_fm_count_dev(ctx.handle, fm._h, C.c_void_p(sfx.data_ptr()), C.c_void_p(so.data_ptr()), len(sub), C.c_void_p(o.data_ptr())) == 0
        z = o.cpu().numpy() == 0
        last[alive & z] = L; alive &= ~z
        if not alive.any(): break
    steps_miss_mean = float(last.mean())
steps = steps_hit + steps_miss_mean * len(miss_idx)
A = steps * 128 + npat * m
print("config 4: n=%d npat=%d m=%d | index build (host text in, incl. suffix sort) %.1f ms" % (n, npat, m, t_build * 1e3))
print("count: best %.2f ms, mean %.2f ms over %d calls -> %.1f Mpatterns/s" % (best * 1e3, mean * 1e3, len(ts), npat / best / 1e6))
print("found %d (%.2f%%); executed steps %.4g (misses stop after %.1f steps on average)" % (hits, 100.0 * hits / npat, steps, steps_miss_mean))
print("A_cnt = steps x 2 x 64 B + pattern bytes = %.1f GB -> %.0f GB/s algorithmic at the best time" % (A / 1e9, A / best / 1e9))
# spot check against naive counting on a few patterns
tb = text.tobytes()
for j in (0, 1, 99, 12345):
    p = pats[j].cpu().numpy().tobytes()
    c, k = 0, tb.find(p)
    while k >= 0:
        c += 1; k = tb.find(p, k + 1)
    assert c == int(out[j]), (j, c, int(out[j]))
print("spot check vs naive substring count ok")
