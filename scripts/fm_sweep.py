"""FM-count rate against the text size (index size): where are the knees?  Same batch shape as BASELINE configs[3]
(100-byte patterns, 99 % substrings) at every size; patterns and index resident in HBM."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import torch, textcomp
from textcomp.synth import c4_patterns_dev
npat = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
ctx = textcomp.Context(0); lib = ctx.lib
lib.tc_fm_count_dev.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p]
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [20, 22, 24, 25, 26, 27, 28, 29, 30]
for lg in sizes:
    n = 1 << lg
    d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert lib.tc_generate_dev(ctx.handle, 0, 0xC4, n, C.c_void_p(d_text.data_ptr())) == 0
    torch.cuda.synchronize()
    fm = ctx.fm_build_dev(d_text)
    pats, d_offs = c4_patterns_dev(ctx, d_text, npat, 100)
    d_out = torch.zeros(npat, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ts = []
    for it in range(4):
        t0 = time.perf_counter()
        assert lib.tc_fm_count_dev(ctx.handle, fm._h, C.c_void_p(pats.data_ptr()), C.c_void_p(d_offs.data_ptr()), npat, C.c_void_p(d_out.data_ptr())) == 0
        if it: ts.append(time.perf_counter() - t0)
    print("text 2^%d: %d patterns in %.2f ms = %.0f Mpat/s = %.1f G steps/s" % (lg, npat, min(ts) * 1e3, npat / min(ts) / 1e6, npat * 99.1 / min(ts) / 1e9), flush=True)
    fm.close(); del d_text, pats, d_offs, d_out
