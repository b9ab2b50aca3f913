"""Multi-GPU leg of the path: independent records, one per rank, no data-path
collective during the encode; ONE exchange per record at the end -- the variable-size
gather of the encoded blocks on rank 0.

Design for xGMI (point-to-point links, no switch): a root gather posted as one batch
of sends/receives lets rank 0 ingest on all of its links at once, where a ring would be
bound by a single link.  Two further measures keep the exchange off the critical path:
  * the payload is the block's CONTAINER of include/textcomp.h (header + packed runs; an ACGTN
    record: a nibble stream of ~0.53 bytes per run instead of 6), and
  * the exchange is pipelined: `submit()` only posts the transfers (double-buffered),
    so the gather of record k overlaps the encode of record k+1; `drain()` completes
    everything still in flight.
Headers travel on their own process group (its own RCCL communicator) so that the small
synchronous header all-gather never queues behind a payload still in flight.

torch.distributed is plumbing here (backend "nccl" is RCCL on ROCm; "gloo" in the CPU
tests)."""
import torch
import torch.distributed as dist

HDR_WORDS = 6  # nbytes, nruns, nesc, primary, sigma, n


class BlockGather:
    def __init__(self, cap_bytes, device, depth=2, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device
        self.cap = int(cap_bytes)
        self.depth = depth
        self.hdr_group = dist.new_group(ranks=list(range(self.world))) if self.world > 1 else None
        self._recv = None     # rank 0: [slot][peer] uint8 buffers
        self._inflight = [None] * depth   # per slot: (works, result)
        self._step = 0
        self.completed = []   # rank 0: finished gathers, oldest first (bounded)
        self.wait_ms = []     # host time spent waiting for posted transfers, per completed slot

    def _alloc(self):
        if self._recv is None:
            self._recv = [[None] + [torch.empty(self.cap, dtype=torch.uint8, device=self.device)
                                    for _ in range(1, self.world)] for _ in range(self.depth)]

    def _finish_slot(self, slot):
        entry = self._inflight[slot]
        if entry is None:
            return
        works, result = entry
        import time
        t0 = time.perf_counter()
        for w in works:
            w.wait()
        if works and self.device.type == "cuda":
            # for RCCL, wait() only orders torch's current stream behind the transfer; the
            # encoder runs on its own HIP stream, so complete the transfer on the host before
            # the caller may overwrite the payload buffer
            torch.cuda.current_stream(self.device).synchronize()
        self.wait_ms.append((time.perf_counter() - t0) * 1e3)
        self._inflight[slot] = None
        if result is not None:
            self.completed.append(result)
            if len(self.completed) > 2 * self.depth:
                self.completed.pop(0)

    def acquire(self):
        """Slot of the NEXT submit, with everything that was posted from it completed: the caller owns
        `depth` payload buffers, writes record k into buffer acquire() and then submits it.  The send
        of record k - depth (same buffer) has finished when this returns, so the buffer may be
        overwritten; on rank 0 the result retained for that record aliases the buffer and is dropped
        from `completed` by then (only the last `depth` results stay valid)."""
        slot = self._step % self.depth
        self._finish_slot(slot)
        while len(self.completed) > self.depth - 1:
            self.completed.pop(0)
        return slot

    def submit(self, header, payload):
        """header: 6 ints (nbytes, nruns, nesc, primary, sigma, n); payload: uint8 tensor holding
        at least header[0] bytes -- the buffer of slot acquire(), which the caller must not touch
        again before acquire() hands the slot back (or drain()).  Returns immediately after posting.
        An oversize payload raises on EVERY rank (all of them see all headers), after the header
        exchange, so no rank is left waiting in a collective."""
        slot = self._step % self.depth
        self._step += 1
        self._finish_slot(slot)
        nbytes = int(header[0])
        if self.world == 1:
            if nbytes > self.cap:
                raise ValueError("packed block of %d bytes exceeds the gather capacity %d" % (nbytes, self.cap))
            self.completed.append([(tuple(int(v) for v in header), payload[:nbytes])])
            if len(self.completed) > 2 * self.depth:
                self.completed.pop(0)
            return
        hdr = torch.tensor([int(v) for v in header], dtype=torch.int64, device=self.device)
        hdrs = torch.empty(self.world * HDR_WORDS, dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(hdrs, hdr, group=self.hdr_group)
        ops, result = [], None
        H = hdrs.view(self.world, HDR_WORDS).tolist()     # one host sync per record (46 ms steps: negligible)
        over = [(r, int(H[r][0])) for r in range(self.world) if int(H[r][0]) > self.cap]
        if over:
            raise ValueError("packed block(s) exceed the gather capacity %d: (rank, bytes) = %s" % (self.cap, over))
        if self.rank == 0:
            self._alloc()
            result = [(tuple(int(v) for v in H[0]), payload[:nbytes])]
            for r in range(1, self.world):
                k = int(H[r][0])
                buf = self._recv[slot][r][:k]
                result.append((tuple(int(v) for v in H[r]), buf))
                if k:
                    ops.append(dist.P2POp(dist.irecv, buf, r, group=self.group))
        elif nbytes:
            ops.append(dist.P2POp(dist.isend, payload[:nbytes], 0, group=self.group))
        works = dist.batch_isend_irecv(ops) if ops else []
        self._inflight[slot] = (works, result)

    def prime(self):
        """Untimed set-up: one tiny exchange so that both communicators and the peer-to-peer
        connections to rank 0 exist before the first real record is posted."""
        if self.world == 1:
            return
        self.submit([16, 0, 0, 0, 0, 0], torch.zeros(16, dtype=torch.uint8, device=self.device))
        self.drain()
        self.completed.clear()
        self._step = 0

    def drain(self):
        """Complete every transfer still in flight (oldest first)."""
        for i in range(self.depth):
            self._finish_slot((self._step + i) % self.depth)


def shard_patterns(npat, world, rank):
    """FM-count shards by pattern batch, index replicated per GPU: contiguous slice of
    the pattern list for this rank (result order = pattern order after concatenation)."""
    per = (npat + world - 1) // world
    lo = min(rank * per, npat)
    return lo, min(lo + per, npat)


class NativeGather:
    """Same contract as BlockGather (acquire / submit / drain / completed), with the exchange done by the
    library's own RCCL communicator behind the C ABI (tc_comm_*: what a Haskell or C caller would use).
    torch.distributed only carries the 128-byte communicator id from rank 0 to the others, once.
    One exchange is in flight at a time; it overlaps the encode of the next record."""

    def __init__(self, ctx, cap_bytes, device, depth=2, group=None):
        import ctypes as C
        from . import _lib
        self._C, self.ctx, self.lib = C, ctx, ctx.lib
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device, self.cap, self.depth = device, (int(cap_bytes) + 255) & ~255, depth
        idbuf = (C.c_uint8 * _lib.TC_COMM_ID_BYTES)()
        rc0 = self.lib.tc_comm_unique_id(ctx.handle, idbuf) if self.rank == 0 else 0
        if self.world > 1:
            # (a failure on rank 0 -- no RCCL to bind -- travels in place of the id: nobody is left waiting)
            box = [bytes(idbuf) if rc0 == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            if box[0] is None:
                raise RuntimeError("tc_comm_unique_id failed on rank 0: %s" % self.lib.tc_last_error(ctx.handle).decode()
                                   if self.rank == 0 else "tc_comm_unique_id failed on rank 0")
            idbuf = (C.c_uint8 * _lib.TC_COMM_ID_BYTES).from_buffer_copy(box[0])
        else:
            ctx._check(rc0)
        h = C.c_void_p()
        ctx._check(self.lib.tc_comm_create(ctx.handle, idbuf, self.rank, self.world, C.byref(h)))
        self._h = h
        self.comm_cus = int(self.lib.tc_comm_reserved_cus(h))
        self._recv = ([torch.empty(self.world * self.cap, dtype=torch.uint8, device=device) for _ in range(depth)]
                      if self.rank == 0 else [None] * depth)
        self._pending = None    # (slot, sizes)
        self._step = 0
        self.completed = []
        self.wait_ms = []       # host time inside tc_comm_wait, per completed exchange

    def _finish(self):
        if self._pending is None:
            return
        slot, sizes = self._pending
        import time
        t0 = time.perf_counter()
        self.ctx._check(self.lib.tc_comm_wait(self._h))
        self.wait_ms.append((time.perf_counter() - t0) * 1e3)
        self._pending = None
        if self.rank == 0:
            buf = self._recv[slot]
            self.completed.append([((int(sizes[r]),), buf[r * self.cap:r * self.cap + int(sizes[r])]) for r in range(self.world)])
            while len(self.completed) > self.depth:
                self.completed.pop(0)

    def acquire(self):
        self._finish()      # one exchange in flight: the caller's buffers and ours are free again
        return self._step % self.depth

    def submit(self, header, payload):
        C = self._C
        self._finish()
        slot = self._step % self.depth
        self._step += 1
        sizes = (C.c_uint64 * self.world)()
        recv = self._recv[slot]
        rc = self.lib.tc_comm_gather(self._h, 0, C.c_void_p(payload.data_ptr()), int(header[0]),
                                     C.c_void_p(recv.data_ptr()) if recv is not None else None, self.cap, sizes)
        if rc == -2:
            raise ValueError("packed block(s) exceed the gather capacity %d: sizes = %s" % (self.cap, list(sizes)))
        self.ctx._check(rc)
        self._pending = (slot, list(sizes))

    def prime(self):
        """one small exchange, completed: RCCL builds its peer-to-peer connections on first use, which must not
        happen inside a timed step"""
        if self.world == 1:
            return
        C = self._C
        tiny = torch.zeros(256, dtype=torch.uint8, device=self.device)
        torch.cuda.synchronize(self.device)     # (the library posts on its own streams, not torch's)
        sizes = (C.c_uint64 * self.world)()
        recv = self._recv[0]
        self.ctx._check(self.lib.tc_comm_gather(self._h, 0, C.c_void_p(tiny.data_ptr()), 256,
                                                C.c_void_p(recv.data_ptr()) if recv is not None else None, self.cap, sizes))
        self.ctx._check(self.lib.tc_comm_wait(self._h))

    def drain(self):
        self._finish()

    def broadcast(self, buf, root=0):
        self.ctx._check(self.lib.tc_comm_broadcast(self._h, root, self._C.c_void_p(buf.data_ptr()), buf.numel()))

    def close(self):
        if getattr(self, "_h", None):
            self._finish()
            self.lib.tc_comm_destroy(self._h)
            self._h = None
