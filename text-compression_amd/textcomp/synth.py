"""Synthetic workloads of SURVEY.md 8(d) that are not a plain record (those come from
tc_generate_dev): the pattern batch of BASELINE configs[3] -- 99 % 100-byte substrings of the text
at offsets splitmix64(0xC4F0, j) mod (n - 99), every 100th pattern iid ACGTN (the miss path)."""
import ctypes as C

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)


def splitmix64_at(seed, i):
    """counter-based splitmix64 (the generator of tc_generate_dev): numpy uint64 arrays"""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (i.astype(np.uint64) + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def c4_offsets(n, npat, m=100):
    return (splitmix64_at(0xC4F0, np.arange(npat, dtype=np.uint64)) % np.uint64(n - m + 1)).astype(np.int64)


def c4_patterns_dev(ctx, d_text, npat, m=100):
    """-> (pats u8[npat, m], offs u64[npat + 1]) on the device of d_text (a torch uint8 tensor)."""
    import torch
    n = d_text.numel()
    dev = d_text.device
    offs = torch.from_numpy(c4_offsets(n, npat, m)).to(dev)
    pats = torch.empty((npat, m), dtype=torch.uint8, device=dev)
    ar = torch.arange(m, device=dev)[None, :]
    for lo in range(0, npat, 1 << 20):       # bounded index tensors
        hi = min(lo + (1 << 20), npat)
        pats[lo:hi] = d_text[(offs[lo:hi, None] + ar).reshape(-1)].reshape(hi - lo, m)
    miss = torch.arange(99, npat, 100, device=dev)
    if len(miss):
        d_rand = torch.empty(len(miss) * m, dtype=torch.uint8, device=dev)
        # the library writes d_rand on ITS stream: torch's kernels above must be done first (the block may
        # be recycled from a temporary that a queued gather still reads)
        torch.cuda.synchronize()
        rc = ctx.lib.tc_generate_dev(ctx.handle, 0, 0xC4F1, len(miss) * m, C.c_void_p(d_rand.data_ptr()))
        assert rc == 0
        torch.cuda.synchronize()
        pats[miss] = d_rand.reshape(-1, m)
    d_offs = (torch.arange(npat + 1, device=dev, dtype=torch.int64) * m).contiguous()
    return pats.contiguous(), d_offs
