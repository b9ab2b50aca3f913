"""CPU leg of BASELINE configs[3] (test infrastructure; run by hand on the GPU box, output kept in
profiles/r02_fm_count.txt).  BASELINE.md 3.2: the reference's only parallel code is the map of
countFMIndex over the pattern list (bytestringFMIndexCountP, FMIndex.hs:411-432, parListChunk
:417-423); GHC is absent, so the port (oracle/tc_oracle.c: orc_fm_count_batch) is timed on 1 thread
and on all host cores, on a bounded sample of the same 10^7-pattern batch against the index of the
same 2^28-byte text.  The sample's counts are then compared with the device's: parity at the FULL
text size (the device index of 2^28 symbols against the oracle's)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
import oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 28)
npat = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
sample = int(sys.argv[3]) if len(sys.argv) > 3 else 400_000
m = 100
cores = os.cpu_count()

import torch  # noqa: E402
import textcomp  # noqa: E402
from textcomp.synth import c4_patterns_dev  # noqa: E402

ctx = textcomp.Context(0)
d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
assert ctx.lib.tc_generate_dev(ctx.handle, 0, 0xC4, n, C.c_void_p(d_text.data_ptr())) == 0
torch.cuda.synchronize()
text = d_text.cpu().numpy()
pats, d_offs = c4_patterns_dev(ctx, d_text, npat, m)
fm = ctx.fm_build(text)
d_out = torch.zeros(npat, dtype=torch.int64, device="cuda")
ctx.lib.tc_fm_count_dev.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p]
torch.cuda.synchronize()   # the library runs on its own stream
assert ctx.lib.tc_fm_count_dev(ctx.handle, fm._h, C.c_void_p(pats.data_ptr()), C.c_void_p(d_offs.data_ptr()), npat,
                               C.c_void_p(d_out.data_ptr())) == 0
dev = d_out.cpu().numpy()
print("device: counted %d patterns against the 2^%d-byte text" % (npat, n.bit_length() - 1), flush=True)

t0 = time.perf_counter()
ofm = O.FMIndex(text)
print("oracle index build (suffix sort + checkpoints, 1 thread): %.1f s" % (time.perf_counter() - t0), flush=True)
flat = pats[:sample].cpu().numpy().reshape(-1)
offs = np.arange(sample + 1, dtype=np.int64) * m
one = min(sample, 40_000)
t0 = time.perf_counter()
w1 = ofm.count_batch(flat[:one * m], offs[:one + 1], threads=1)
t1 = time.perf_counter() - t0
print("cpu port, 1 thread : %d patterns in %.2f s = %.1f kpatterns/s" % (one, t1, one / t1 / 1e3), flush=True)
t0 = time.perf_counter()
wall = ofm.count_batch(flat, offs, threads=cores)
ta = time.perf_counter() - t0
print("cpu port, %d threads (all host cores): first %d patterns of the batch in %.2f s = %.1f kpatterns/s"
      % (cores, sample, ta, sample / ta / 1e3), flush=True)
assert np.array_equal(w1, wall[:one])
assert np.array_equal(wall, dev[:sample]), "device counts differ from the oracle's at full text size"
print("parity: device counts == oracle counts on the first %d patterns (text 2^%d)" % (sample, n.bit_length() - 1))
