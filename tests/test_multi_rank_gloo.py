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


def _block_for(rank, step, n):
    rng = np.random.default_rng(1000 + 17 * rank + step)
    k = int(n * (0.5 + 0.1 * rank)) + step
    hdr = [k, k - 3, 0, 17 + rank, 6, n]
    return hdr, torch.from_numpy(rng.integers(0, 256, n + 64).astype(np.uint8))


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from textcomp.gather import BlockGather, shard_patterns
        g = BlockGather(n + 64, torch.device("cpu"), depth=2)
        g.prime()
        ok = g.completed == [] and g._step == 0
        steps = 5
        # the caller owns `depth` payload buffers and REUSES them: acquire() hands a slot back only when
        # whatever was posted from it has completed (a send still in flight must not be overwritten)
        bufs = [torch.zeros(n + 64, dtype=torch.uint8) for _ in range(2)]
        for st in range(steps):     # pipelined: submit only posts
            hdr, payload = _block_for(rank, st, n)
            slot = g.acquire()
            ok &= slot == st % 2
            bufs[slot].copy_(payload)
            g.submit(hdr, bufs[slot])
        g.drain()
        if rank == 0:
            got = g.completed[-2:]          # the last `depth` records are still retained (their buffers untouched)
            for st, res in zip(range(steps - 2, steps), got):
                for r in range(world):
                    ehdr, epay = _block_for(r, st, n)
                    hdr, data = res[r]
                    ok &= list(hdr) == ehdr
                    ok &= torch.equal(data, epay[:ehdr[0]])
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


def _worker_oversize(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from textcomp.gather import BlockGather
        g = BlockGather(1000, torch.device("cpu"), depth=2)
        g.prime()
        raised = False
        try:   # only rank 1's payload is too large: EVERY rank must raise (nobody is left in a collective)
            k = 2000 if rank == 1 else 500
            g.submit([k, 0, 0, 0, 6, 100], torch.zeros(2048, dtype=torch.uint8))
        except ValueError:
            raised = True
        q.put((rank, raised))
    finally:
        dist.destroy_process_group()


def test_oversize_payload_raises_on_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_oversize, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


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
