import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import textcomp
ctx = textcomp.Context(0)
f = ctx.lib.tc_dbg_stream_bench
f.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
for mode, name in ((0, "copy"), (1, "read"), (2, "write")):
    for w in (16, 8, 4, 2, 1):
        g = C.c_double()
        rc = f(ctx.handle, 4 << 30, w, mode, 5, C.byref(g))
        print("%-5s width %2d B/lane: %8.1f GB/s (rc=%d)" % (name, w, g.value, rc), flush=True)
