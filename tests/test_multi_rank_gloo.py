"""world_size-2 gloo rehearsal (CPU) of the multi-GPU leg: header all-gather +
variable-size gather of encoded blocks on rank 0, and pattern sharding for FM-count."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _block_for(rank, n):
    rng = np.random.default_rng(1000 + rank)
    k = int(n * (0.5 + 0.1 * rank))
    return (k, 17 + rank, 6, n, torch.from_numpy(rng.integers(1, 9, n + 2).astype(np.int32)),
            torch.from_numpy(rng.integers(0, 6, n + 2).astype(np.int16)))


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from textcomp.gather import BlockGather, shard_patterns
        k, prim, sig, nn, cnt, val = _block_for(rank, n)
        g = BlockGather(n + 2, torch.device("cpu"))
        ok = True
        for _ in range(2):  # buffers are reused across steps
            res = g.gather(k, prim, sig, nn, cnt, val)
            if rank == 0:
                for r in range(world):
                    ek, eprim, esig, en, ecnt, eval_ = _block_for(r, n)
                    hdr, c, v = res[r]
                    ok &= hdr == (ek, eprim, esig, en)
                    ok &= torch.equal(c, ecnt[:ek]) and torch.equal(v, eval_[:ek])
            else:
                ok &= res is None
        lo, hi = shard_patterns(11, world, rank)
        part = torch.arange(lo, hi, dtype=torch.int64)
        parts = [torch.empty(6, dtype=torch.int64) for _ in range(world)]
        pad = torch.full((6,), -1, dtype=torch.int64)
        pad[:hi - lo] = part
        dist.all_gather(parts, pad)
        merged = torch.cat([p[p >= 0] for p in parts])
        ok &= merged.tolist() == list(range(11))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_gather_world2():
    import sys
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 5000, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]
