"""What the memory system gives the access pattern of a radix pass (no ranking, no look-back, no
LDS): tiles of 4096 (u64, u32) pairs read coalesced, written as `bins` segments per tile."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import textcomp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
ctx = textcomp.Context(0)
f = ctx.lib.tc_dbg_scatter_bench
f.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_double)]
for bins, xrun in ((1, 0), (16, 0), (64, 0), (128, 0), (256, 0), (1024, 0), (125, 0), (125, 1), (125, 2), (125, 4), (125, 8), (125, 16), (125, 32), (250, 8), (63, 8), (31, 8), (15, 8), (125, 8 + 256), (125, 8 + 512)):
    ms = C.c_double()
    rc = f(ctx.handle, n, bins, xrun, 6, C.byref(ms))
    print("bins %4d xcd-run %2d: %.3f ms per pass = %.0f GB/s (24 B per pair)  rc=%d" % (bins, xrun, ms.value, 24 * n / (ms.value * 1e-3) / 1e9 if ms.value else 0, rc), flush=True)
