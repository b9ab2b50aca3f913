"""Does a partition by the top bits of the target index make the rank scatter / gather cheaper?
(random 4-byte accesses over 4 GiB against the same accesses grouped into 2^k regions)"""
import sys, torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(1)
idx = torch.randperm(n, device=dev, generator=g)
vals = torch.arange(n, device=dev, dtype=torch.int32)
isa = torch.zeros(n, device=dev, dtype=torch.int32)
def t(f, reps=3):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("n = %d" % n)
print("random scatter  %.2f ms" % t(lambda: isa.index_copy_(0, idx, vals)), flush=True)
print("random gather   %.2f ms" % t(lambda: torch.index_select(isa, 0, idx)), flush=True)
for bits in (4, 6, 8, 10, 12):
    sh = (n.bit_length() - 1) - bits
    key = (idx >> sh).to(torch.int16)
    order = torch.sort(key, stable=True).indices
    pidx = idx[order]
    del key, order
    print("regions 2^%-2d (%6.1f MiB of ranks each): scatter %.2f ms  gather %.2f ms" % (
        bits, 4.0 * (1 << sh) / 2**20, t(lambda: isa.index_copy_(0, pidx, vals)),
        t(lambda: torch.index_select(isa, 0, pidx))), flush=True)
    del pidx
