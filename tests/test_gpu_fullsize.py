"""BASELINE configs[2] at full size (1 GiB ACGTN record, device resident) and the largest record
the ABI admits (TC_MAX_N = 2^31 - 16 bytes, ~130 GB of workspace): the oracle cannot run there,
so parity is checked through size-independent properties -- the decode of the encode is the
input, bit for bit; the run lengths sum to N; exactly one primary row."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [(1 << 30), 0x7ffffff0])
def test_full_size_roundtrip(n):
    import torch
    import textcomp
    from textcomp import Block
    if n > (1 << 30) and torch.cuda.mem_get_info()[0] < 190 * (1 << 30):
        pytest.skip("needs ~150 GB of free HBM")
    ctx = textcomp.Context(0)
    lib = ctx.lib
    d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert lib.tc_generate_dev(ctx.handle, 0, 0xC3, n, C.c_void_p(d_text.data_ptr())) == 0
    cap = n + 2
    d_cnt = torch.empty(cap, dtype=torch.int32, device="cuda")
    d_val = torch.empty(cap, dtype=torch.int16, device="cuda")
    blk = Block()
    blk.nruns = cap
    blk.run_count = d_cnt.data_ptr()
    blk.run_value = d_val.data_ptr()
    rc = lib.tc_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk))
    assert rc == 0, lib.tc_last_error(ctx.handle)
    k = int(blk.nruns)
    assert int(blk.n) == n and int(blk.sigma) == 6 and 0 < int(blk.primary) <= n
    assert int(d_cnt[:k].sum(dtype=torch.int64).item()) == n + 1          # runs cover the BWT exactly
    assert int(d_val[:k].max().item()) < 6
    assert 0.75 < k / (n + 1) < 0.85                                      # iid ACGTN: ~0.8 N runs
    st = ctx.stats()
    assert st.finish_pass == 1 and st.rounds <= 3
    d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    rc = lib.tc_decode_dev(ctx.handle, C.byref(blk), C.c_void_p(d_out.data_ptr()))
    assert rc == 0, lib.tc_last_error(ctx.handle)
    assert bool(torch.equal(d_out, d_text))
    ctx.close()
