"""Multi-GPU FM-index count (last sentence): the index is built once, replicated on every
GPU by ONE broadcast, the pattern batch is cut into contiguous slices (one per rank), every rank
counts its slice with the single-GPU kernel, and the counts are gathered in pattern order.  The only
parallelism of the reference this replaces is parListChunk over the pattern list
(FMIndex.hs:417-423, bytestringFMIndexCountP :411-432): same values, same order.

torch.distributed is plumbing (backend "nccl" = RCCL on ROCm; "gloo" in the one-GPU rehearsal, where the
payload is staged through the host)."""
import torch
import torch.distributed as dist

from . import FMIndexHandle
from .gather import shard_patterns


def replicate_index(ctx, fm, src=0, group=None, with_locate=False):
    """Rank `src` holds `fm` (others pass None); returns this rank's copy (rank src: fm itself).
    One size broadcast + one payload broadcast."""
    rank = dist.get_rank(group)
    host = dist.get_backend(group) == "gloo"
    dev = torch.device("cuda", ctx.device)
    xdev = torch.device("cpu") if host else dev
    meta = torch.zeros(2, dtype=torch.int64, device=xdev)
    payload = None
    if rank == src:
        payload = fm.export_dev(with_locate)
        meta[0], meta[1] = payload.numel(), fm.n
    dist.broadcast(meta, src, group=group)
    nbytes, n = int(meta[0].item()), int(meta[1].item())
    if rank == src:
        wire = payload.cpu() if host else payload
    else:
        wire = torch.empty(nbytes, dtype=torch.uint8, device=xdev)
    dist.broadcast(wire, src, group=group)
    if rank == src:
        return fm
    return FMIndexHandle.import_dev(ctx, wire.to(dev) if host else wire, n)


def sharded_count(ctx, fm, d_pats, d_offs, npat, group=None):
    """Every rank holds the whole batch description (d_pats flat uint8, d_offs int64 [npat + 1], on its
    device) and a copy of the index; rank r counts patterns shard_patterns(npat, world, r).  Returns the
    int64 counts of ALL patterns, in pattern order, on every rank (0 stands for Nothing)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    host = dist.get_backend(group) == "gloo"
    lo, hi = shard_patterns(npat, world, rank)
    per = (npat + world - 1) // world
    mine = torch.zeros(per, dtype=torch.int64, device=d_pats.device)
    if hi > lo:
        b0 = int(d_offs[lo].item())
        offs = (d_offs[lo:hi + 1] - b0).contiguous()
        pats = d_pats[b0:int(d_offs[hi].item())]
        if pats.numel() == 0:
            pats = torch.zeros(16, dtype=torch.uint8, device=d_pats.device)
        mine[:hi - lo] = fm.count_dev(pats.contiguous(), offs, hi - lo)
    torch.cuda.synchronize()
    allc = torch.empty(per * world, dtype=torch.int64, device="cpu" if host else d_pats.device)
    dist.all_gather_into_tensor(allc, mine.cpu() if host else mine, group=group)
    allc = allc.to(d_pats.device)
    parts = []
    for r in range(world):
        a, b = shard_patterns(npat, world, r)
        parts.append(allc[r * per:r * per + (b - a)])
    return torch.cat(parts)
