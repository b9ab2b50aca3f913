"""BASELINE configs[3] at FULL size: FM-index count of 10^7 x 100-byte ACGTN patterns against the
index of a 2^28-byte text (countFMIndex, FMIndex/Internal.hs:347-438, mapped over the batch as
bytestringFMIndexCountP does, FMIndex.hs:411-432).  At this size the oracle's index is minutes of
CPU, so the test checks (i) the generator's promise -- every substring pattern is found, every
iid pattern is not --, (ii) naive substring counting on a sample of patterns over the whole text,
and (iii) the oracle itself on the same workload shape over a 2^24-byte text (10^5 patterns).
tests/long/fm_cpu_port.py compares a sample with the oracle's index of the full 2^28 text."""
import ctypes as C

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


def _count_dev(ctx, fm, pats, d_offs, npat):
    import torch
    d_out = torch.zeros(npat, dtype=torch.int64, device="cuda")
    ctx.lib.tc_fm_count_dev.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p]
    torch.cuda.synchronize()   # the library runs on its own stream
    rc = ctx.lib.tc_fm_count_dev(ctx.handle, fm._h, C.c_void_p(pats.data_ptr()), C.c_void_p(d_offs.data_ptr()),
                                 npat, C.c_void_p(d_out.data_ptr()))
    assert rc == 0, ctx.lib.tc_last_error(ctx.handle)
    return d_out.cpu().numpy()


def _workload(ctx, n, npat):
    import torch
    from textcomp.synth import c4_patterns_dev
    d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert ctx.lib.tc_generate_dev(ctx.handle, 0, 0xC4, n, C.c_void_p(d_text.data_ptr())) == 0
    torch.cuda.synchronize()
    text = d_text.cpu().numpy()
    pats, d_offs = c4_patterns_dev(ctx, d_text, npat)
    return text, pats, d_offs


def test_config4_full_size():
    import textcomp
    n, npat = 1 << 28, 10_000_000
    ctx = textcomp.Context(0)
    text, pats, d_offs = _workload(ctx, n, npat)
    fm = ctx.fm_build(text)
    out = _count_dev(ctx, fm, pats, d_offs, npat)
    is_miss = (np.arange(npat) % 100) == 99
    assert (out[~is_miss] >= 1).all()            # a substring of the text occurs in it
    assert (out[is_miss] == 0).all()             # 100 iid symbols: 5^-100 per position
    assert int((out > 1).sum()) < npat // 1000   # 100-byte windows of an iid text do not repeat
    tb = text.tobytes()
    sample = list(range(0, npat, npat // 24))[:24] + [99, 199, 9_999_999]
    hp = pats[sample].cpu().numpy()
    for row, j in zip(hp, sample):
        p, c, k = row.tobytes(), 0, 0
        k = tb.find(p)
        while k >= 0:
            c += 1
            k = tb.find(p, k + 1)
        assert c == int(out[j]), (j, c, int(out[j]))
    fm.close()
    ctx.close()


def test_config4_shape_against_oracle_16m():
    import textcomp
    n, npat = 1 << 24, 100_000
    ctx = textcomp.Context(0)
    text, pats, d_offs = _workload(ctx, n, npat)
    assert np.array_equal(text, O.gen_acgtn(0xC4, n))
    fm = ctx.fm_build(text)
    out = _count_dev(ctx, fm, pats, d_offs, npat)
    ofm = O.FMIndex(text)
    want = ofm.count_batch(pats.cpu().numpy().reshape(-1), d_offs.cpu().numpy(), threads=4)
    assert np.array_equal(out, want)
    assert int((want == 0).sum()) == npat // 100
    fm.close()
    ctx.close()
