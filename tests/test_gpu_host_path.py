"""The host-pointer entry points -- what a Haskell caller reaches (bytestringToBWT, bytestringToBWTToRLEB: reference
BWT.hs:68-70, RLE.hs:83-85 take a host ByteString).  Round 4: the copies go through the context's page-locked staging
ring (helper threads, pieces of 16 MiB) unless the caller's buffer is page-locked itself, and the device-side buffers
stay with the context.  Checked here: sizes that are not multiples of a piece, both directions, pageable and page-locked
buffers, growth and reuse across calls -- each against the device-pointer path (tested against the oracle elsewhere) and,
at a size the oracle does in seconds, against the oracle itself."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-compression_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import textcomp
    with textcomp.Context(0) as c:
        yield c


def _dev_container(ctx, text):
    import torch
    n = len(text)
    d = torch.from_numpy(text).cuda()
    cap = int(ctx.lib.tc_container_bound(n + 2, 257))
    out = torch.empty(cap + 16, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    used = ctx.encode_container_dev(d.data_ptr(), n, out.data_ptr(), cap)
    return out[:used].cpu().numpy().tobytes()


@pytest.mark.parametrize("n", [(1 << 20) + 17, (16 << 20) - 1, (16 << 20) + 1, (37 << 20) + 123, (3 << 20), (80 << 20) + 5])
def test_pageable_host_buffers_equal_the_device_path(ctx, n):
    import oracle as O
    text = O.gen_acgtn(0x4057 + n, n)
    blob = ctx.encode_container(text)          # numpy (pageable) buffers in and out: the staging ring, both ways
    assert blob == _dev_container(ctx, text)
    assert ctx.decode_container(blob) == text.tobytes()


def test_page_locked_host_buffers(ctx):
    import torch
    import oracle as O
    n = (33 << 20) + 7
    text = O.gen_acgtn(0x9137, n)
    pt = torch.from_numpy(text).pin_memory()
    cap = int(ctx.lib.tc_container_bound(n + 2, 257))
    po = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
    used = C.c_uint64(cap)
    ctx.lib.tc_encode_container.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
    rc = ctx.lib.tc_encode_container(ctx.handle, C.c_void_p(pt.data_ptr()), n, C.c_void_p(po.data_ptr()), C.byref(used))
    assert rc == 0, ctx.lib.tc_last_error(ctx.handle)
    assert po[:used.value].numpy().tobytes() == _dev_container(ctx, text)


def test_run_arrays_to_host_against_the_oracle(ctx):
    import oracle as O
    n = (5 << 20) + 3                           # the 4.8 bytes per input byte of raw runs: 25 MB through the ring
    text = O.gen_acgtn(0x51, n)
    L = O.bwt_encode_arr(text)
    idx, fl = O.mtf_encode_arr(L)
    counts, vals = O.rle_encode_u32_arr(idx)
    blk = ctx.encode(text)
    assert blk["primary"] == int(np.nonzero(L < 0)[0][0]) and blk["final_list"].tolist() == fl.tolist()
    assert np.array_equal(blk["run_count"], counts) and np.array_equal(blk["run_value"], vals)
    assert ctx.decode(blk) == text.tobytes()


def test_buffers_are_reused_and_grow(ctx):
    import oracle as O
    blobs = {}
    for n in ((2 << 20) + 1, (9 << 20) + 3, (2 << 20) + 1, 1000, (9 << 20) + 3, 0):
        text = O.gen_acgtn(0x77, n)
        b = ctx.encode_container(text)
        assert ctx.decode_container(b) == text.tobytes()
        assert blobs.setdefault(n, b) == b      # the same record gives the same bytes whatever ran before it


def test_direct_copies_when_staging_is_off(ctx):
    import oracle as O
    text = O.gen_acgtn(0x99, (6 << 20) + 11)
    want = ctx.encode_container(text)
    os.environ["TC_HOST_STAGED"] = "0"
    try:
        assert ctx.encode_container(text) == want
    finally:
        os.environ.pop("TC_HOST_STAGED", None)
