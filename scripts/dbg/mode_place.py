"""One process, several workspace placements (tc_ctx_place_workspace with TC_PLACE_ALL=1): the 1 GiB encode time per
placement.  Run plain, or under rocprofv3 --pmc ... --kernel-trace (scripts/mode_pmc.sh joins counters and durations
of msd_partition_kernel per placement)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
os.environ.setdefault("TC_PLACE_ALL", "1")
import torch, textcomp
from textcomp import Block
n = 1 << 30
tries = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = textcomp.Context(0); lib = ctx.lib
d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
cap = n + 2
d_cnt = torch.empty(cap, dtype=torch.int32, device="cuda"); d_val = torch.empty(cap, dtype=torch.int16, device="cuda")
assert lib.tc_generate_dev(ctx.handle, 0, 0xC3, n, C.c_void_p(d_text.data_ptr())) == 0
torch.cuda.synchronize()
blk = Block(); blk.nruns = cap; blk.run_count = d_cnt.data_ptr(); blk.run_value = d_val.data_ptr()
pms, pch = ctx.place_workspace(d_text.data_ptr(), n, blk, tries=tries)
print("PLACEMENTS_MS", " ".join("%.2f" % x for x in pms), "chosen", pch, flush=True)
