"""Multi-GPU leg of the path: independent records, one per rank, no data-path
collective during the encode; ONE exchange at the end -- the variable-size gather of
the encoded blocks on rank 0.

Headers (nruns, primary, sigma, n) travel by all_gather (tiny); the run arrays by
point-to-point sends posted together, so rank 0 receives from all peers at once:
xGMI is point-to-point, a root gather ingests on every link concurrently whereas a
ring would be bound by one link.  torch.distributed is plumbing (backend "nccl" is
RCCL on ROCm; "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


class BlockGather:
    def __init__(self, cap, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device
        self.cap = cap
        self._recv_cnt = None
        self._recv_val = None

    def _alloc(self):
        if self._recv_cnt is None:
            self._recv_cnt = [None] + [torch.empty(self.cap, dtype=torch.int32, device=self.device)
                                       for _ in range(1, self.world)]
            self._recv_val = [None] + [torch.empty(self.cap, dtype=torch.int16, device=self.device)
                                       for _ in range(1, self.world)]

    def gather(self, nruns, primary, sigma, n, run_count, run_value):
        """run_count int32[>=nruns], run_value int16[>=nruns] on self.device.
        Rank 0 returns [(header tuple, counts view, values view)] per rank; others None."""
        hdr = torch.tensor([nruns, primary, sigma, n], dtype=torch.int64, device=self.device)
        hdrs = torch.empty(self.world * 4, dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(hdrs, hdr, group=self.group)
        if self.world == 1:
            return [((nruns, primary, sigma, n), run_count[:nruns], run_value[:nruns])]
        ops = []
        if self.rank == 0:
            self._alloc()
            H = hdrs.view(self.world, 4).tolist()
            for r in range(1, self.world):
                k = int(H[r][0])
                ops.append(dist.P2POp(dist.irecv, self._recv_cnt[r][:k], r, group=self.group))
                ops.append(dist.P2POp(dist.irecv, self._recv_val[r][:k], r, group=self.group))
        else:
            ops.append(dist.P2POp(dist.isend, run_count[:nruns], 0, group=self.group))
            ops.append(dist.P2POp(dist.isend, run_value[:nruns], 0, group=self.group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if self.rank != 0:
            return None
        out = [(tuple(int(v) for v in H[0]), run_count[:nruns], run_value[:nruns])]
        for r in range(1, self.world):
            k = int(H[r][0])
            out.append((tuple(int(v) for v in H[r]), self._recv_cnt[r][:k], self._recv_val[r][:k]))
        return out


def shard_patterns(npat, world, rank):
    """FM-count shards by pattern batch, index replicated per GPU: contiguous slice of
    the pattern list for this rank (result order = pattern order after concatenation)."""
    per = (npat + world - 1) // world
    lo = min(rank * per, npat)
    return lo, min(lo + per, npat)
