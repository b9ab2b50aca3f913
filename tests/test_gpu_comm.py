"""The exchange of the multi-GPU path behind the C ABI (tc_comm_*), as far as one GPU
goes: RCCL is found and bound at run time, a communicator of one rank is built from a unique id, the
container gather (all-gather of the sizes + root copy) and the broadcast run on it, a container larger
than the slot is TC_ERR_CAPACITY.  The point-to-point transfers need a second GPU: the driver's run."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_comm_world1_gather_and_broadcast():
    import torch
    import textcomp
    from textcomp import Block
    from textcomp.gather import NativeGather
    ctx = textcomp.Context(0)
    lib = ctx.lib
    n = 1 << 20
    t = O.gen_acgtn(0xC500, n)
    d_text = torch.from_numpy(t).cuda()
    cap = n + 2
    d_cnt = torch.empty(cap, dtype=torch.int32, device="cuda")
    d_val = torch.empty(cap, dtype=torch.int16, device="cuda")
    blk = Block()
    blk.nruns, blk.run_count, blk.run_value = cap, d_cnt.data_ptr(), d_val.data_ptr()
    torch.cuda.synchronize()
    assert lib.tc_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk)) == 0
    pcap = n + n // 4 + 4096
    packed = torch.empty(pcap, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    nb = C.c_uint64(pcap)
    assert lib.tc_block_to_container_dev(ctx.handle, C.byref(blk), C.c_void_p(packed.data_ptr()), C.byref(nb)) == 0
    g = NativeGather(ctx, pcap, torch.device("cuda", 0), depth=2)
    for step in range(3):       # slots alternate; one exchange in flight
        assert g.acquire() == step % 2
        g.submit([nb.value], packed)
    g.drain()
    (hdr, got), = g.completed[-1]
    assert hdr[0] == nb.value and torch.equal(got, packed[:nb.value])
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    vb = Block()
    vb.nruns, vb.run_count, vb.run_value = cap, d_cnt.data_ptr(), d_val.data_ptr()
    pd = got.clone()
    torch.cuda.synchronize()
    assert lib.tc_container_to_block_dev(ctx.handle, C.c_void_p(pd.data_ptr()), nb.value, C.byref(vb)) == 0
    assert lib.tc_decode_dev(ctx.handle, C.byref(vb), C.c_void_p(back.data_ptr())) == 0
    assert np.array_equal(back.cpu().numpy(), t)
    with pytest.raises(ValueError):     # larger than the slot: refused before anything moves
        small = NativeGather(ctx, 1024, torch.device("cuda", 0))
        small.submit([nb.value], packed)
    x = torch.arange(1000, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    g.broadcast(x)
    assert torch.equal(x, torch.arange(1000, dtype=torch.uint8, device="cuda"))
    g.close()
    ctx.close()


def test_missing_rccl_is_a_status_code_not_a_crash():
    """a host without RCCL: tc_comm_unique_id / tc_comm_create return TC_ERR_NCCL with a message (the binding is
    made at run time; TC_RCCL_LIB names the one library to try).  A child process, because the binding is made once."""
    import subprocess
    import sys
    code = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
import textcomp
ctx = textcomp.Context(0)
ident = (C.c_uint8 * 128)()
rc = ctx.lib.tc_comm_unique_id(ctx.handle, ident)
msg = ctx.lib.tc_last_error(ctx.handle).decode()
h = C.c_void_p()
rc2 = ctx.lib.tc_comm_create(ctx.handle, ident, 0, 1, C.byref(h))
print(rc, rc2, msg)
assert rc == -7 and rc2 == -7 and "librccl not found" in msg, (rc, rc2, msg)
""" % os.path.join(ROOT, "text-compression_amd")
    env = dict(os.environ)
    env["TC_RCCL_LIB"] = "/nonexistent/librccl-none.so"
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr


def test_comm_stream_on_reserved_cus(monkeypatch):
    """TC_COMM_CUS: the communicator's stream is restricted to a few compute units and the context's partition
    levels split their work over the others (what every rank of the N > 1 step runs with).  One rank here: the
    restricted stream carries the size all-gather, the root copy and the broadcast; the record encoded with the
    reduced partition grid (the MSD round 0 forced at this size) gives the container of the plain context."""
    import torch
    import textcomp
    from textcomp.gather import NativeGather
    n = (1 << 21) + 12345
    t = O.gen_acgtn(0xC501, n)
    d_text = torch.from_numpy(t).cuda()
    pcap = n + n // 4 + 4096
    monkeypatch.setenv("TC_SA_MSD_MIN_LOG2", "10")
    with textcomp.Context(0) as plain:
        ref = torch.zeros(pcap, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        nref = plain.encode_container_dev(d_text.data_ptr(), n, ref.data_ptr(), pcap)
        assert plain.stats().msd_path == 1
    monkeypatch.setenv("TC_COMM_CUS", "8")
    ctx = textcomp.Context(0)
    g = NativeGather(ctx, pcap, torch.device("cuda", 0), depth=2)
    assert g.comm_cus == 8
    buf = torch.zeros(pcap, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for step in range(2):
        g.acquire()
        nb = ctx.encode_container_dev(d_text.data_ptr(), n, buf.data_ptr(), pcap)
        assert ctx.stats().msd_path == 1
        g.submit([nb], buf)
    g.drain()
    (hdr, got), = g.completed[-1]
    assert hdr[0] == nref and torch.equal(got, ref[:nref])
    g.close()
    ctx.close()
