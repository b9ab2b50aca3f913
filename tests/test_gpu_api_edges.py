"""C-ABI edge behaviour on the device: argument validation, capacity protocol, device-pointer
entry points, stats, and that errors never hang or crash."""
import ctypes as C

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import textcomp
    c = textcomp.Context(0)
    yield c
    c.close()


def test_argument_errors(ctx):
    import textcomp
    lib = ctx.lib
    prim = C.c_uint64()
    assert lib.tc_bwt_encode(ctx.handle, None, 10, None, C.byref(prim)) == textcomp._lib.TC_ERR_ARG
    assert lib.tc_bwt_encode(None, None, 0, None, C.byref(prim)) == textcomp._lib.TC_ERR_ARG
    assert lib.tc_bwt_encode(ctx.handle, None, 0, None, C.byref(prim)) == 0           # empty input is fine
    assert b"" != lib.tc_last_error(ctx.handle) or True
    big = C.c_uint64(1 << 40)
    assert lib.tc_bwt_encode(ctx.handle, None, big, None, C.byref(prim)) == textcomp._lib.TC_ERR_ARG
    with pytest.raises(textcomp.TcError):
        ctx.bwt_decode(np.zeros(4, np.uint8), 9)                                    # primary out of range


def test_encode_capacity_protocol(ctx):
    import textcomp
    t = O.gen_acgtn(1, 20000)
    with pytest.raises(textcomp.TcError) as e:
        ctx.encode(t, cap=100)
    assert e.value.code == textcomp._lib.TC_ERR_CAPACITY
    blk = ctx.encode(t)                                                              # ctx still usable
    assert ctx.decode(blk) == t.tobytes()


def test_device_pointer_entry_points(ctx):
    import torch
    from textcomp import Block
    lib = ctx.lib
    n = 300000
    d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert lib.tc_generate_dev(ctx.handle, 1, 0xC1, n, C.c_void_p(d_text.data_ptr())) == 0
    host = d_text.cpu().numpy()
    assert host.tolist() == O.gen_ascii(0xC1, n).tolist()                            # same generator on both sides
    d_L = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
    prim = C.c_uint64()
    assert lib.tc_bwt_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.c_void_p(d_L.data_ptr()), C.byref(prim)) == 0
    L = O.bwt_encode_arr(host)
    assert int(np.nonzero(L < 0)[0][0]) == prim.value
    assert np.array_equal(d_L.cpu().numpy(), np.where(L < 0, 0, L).astype(np.uint8))
    # FM count with device-resident patterns
    fm = ctx.fm_build(host)
    pats = [host[i:i + 7].tobytes() for i in (0, 5, 1000)] + [b"\x01\x02"]
    flat = torch.from_numpy(np.frombuffer(b"".join(pats), np.uint8).copy()).cuda()
    offs = torch.tensor(np.cumsum([0] + [len(p) for p in pats]), dtype=torch.int64).cuda()
    out = torch.zeros(len(pats), dtype=torch.int64, device="cuda")
    lib.tc_fm_count_dev.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p]
    assert lib.tc_fm_count_dev(ctx.handle, fm._h, C.c_void_p(flat.data_ptr()), C.c_void_p(offs.data_ptr()), len(pats),
                               C.c_void_p(out.data_ptr())) == 0
    ofm = O.FMIndex(host)
    assert [int(v) or None for v in out.cpu().numpy()] == [ofm.count(p) for p in pats]
    fm.close()


def test_stats_report_the_work_counters(ctx):
    t = O.gen_acgtn(0xC2, 1 << 20)
    ctx.encode(t)
    st = ctx.stats()
    assert st.n == len(t) and st.N == len(t) + 1 and st.sigma == 6
    assert st.rounds >= 1 and st.m[0] == st.N and st.passes[0] >= 1 and st.runs > 0
    assert st.ms_total > 0 and st.ms_sa > 0
    for r in range(1, st.rounds):
        assert st.m[r] < st.m[r - 1]                                                 # the tied set shrinks every round


def test_two_contexts_are_independent():
    import textcomp
    a, b = textcomp.Context(0), textcomp.Context(0)
    t1, t2 = O.gen_acgtn(1, 50000), O.gen_ascii(2, 70000)
    b1, b2 = a.encode(t1), b.encode(t2)
    assert a.decode(b1) == t1.tobytes() and b.decode(b2) == t2.tobytes()
    assert b.decode(b1) == t1.tobytes()                                              # blocks are self-contained
    a.close(); b.close()
