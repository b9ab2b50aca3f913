"""Radix-pass micro-benchmark: ms per pass for the variants / diagnostics given in env."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import textcomp
n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 30)
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 48
check = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = textcomp.Context(0)
f = ctx.lib.tc_dbg_sort_bench
f.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
ms = C.c_double()
rc = f(ctx.handle, n, bits, 2, check, C.byref(ms))
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("TC_"))
print("n=%d bits=%d [%s]: %.3f ms/pass -> %.0f GB/s (24 B/elt)  rc=%d %s" % (
    n, bits, tag, ms.value, 24 * n / (ms.value * 1e-3) / 1e9 if ms.value else 0, rc,
    ctx.lib.tc_last_error(ctx.handle).decode() if rc else ""), flush=True)
