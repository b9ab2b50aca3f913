"""Multi-GPU FM-index count rehearsed on one GPU: the index built on rank 0 is
replicated by one broadcast (tc_fm_export_dev -> broadcast -> tc_fm_import_dev), every rank counts its
contiguous slice of the pattern batch, the counts are gathered in pattern order -- and equal the
single-rank call (bytestringFMIndexCountP returns the same values in the same order as ...CountS,
FMIndex.hs:411-432).  Two ranks share cuda:0 and talk over gloo here; the RCCL run is the 8-GPU job."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import ctypes as C
    import torch
    import torch.distributed as dist
    import textcomp
    from textcomp.fmshard import replicate_index, sharded_count
    from textcomp.synth import c4_patterns_dev
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        ctx = textcomp.Context(0)
        n, npat = 1 << 20, 5001
        d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
        assert ctx.lib.tc_generate_dev(ctx.handle, 0, 0xC4, n, C.c_void_p(d_text.data_ptr())) == 0
        torch.cuda.synchronize()
        pats, d_offs = c4_patterns_dev(ctx, d_text, npat)
        fm = ctx.fm_build(d_text.cpu().numpy()) if rank == 0 else None
        mine = replicate_index(ctx, fm, src=0)
        got = sharded_count(ctx, mine, pats.reshape(-1), d_offs, npat)
        ok = True
        if rank == 0:
            want = fm.count_dev(pats.reshape(-1), d_offs, npat)
            ok = bool(torch.equal(got, want)) and int((want == 0).sum()) == npat // 100
            # an imported count-only index refuses locate instead of faulting
        else:
            try:
                mine.locate([b"ACGT"])
                ok = False
            except textcomp.TcError as e:
                ok = e.code == textcomp._lib.TC_ERR_ARG
        # the full form (with the locate part) answers locate like the original
        full = replicate_index(ctx, fm, src=0, with_locate=True)
        pl = [bytes(pats[j].cpu().numpy()) for j in (0, 1, 2, 99)]
        hits = full.locate(pl)
        ref = [None] * len(pl)
        if rank == 0:
            ref = [h.tolist() for h in fm.locate(pl)]
        obj = [ref]
        dist.broadcast_object_list(obj, src=0)
        ok &= [h.tolist() for h in hits] == obj[0]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_replicated_index_sharded_patterns_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]
