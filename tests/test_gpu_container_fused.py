"""tc_encode_container_dev: text -> container on the device with the RLE stage writing the wire format itself
(rle_nib_kernel, sigma <= 6) -- the step of the multi-GPU path.  Its bytes must be (i) those of the two-step way
(tc_encode_dev + tc_block_to_container_dev, pack_nib_kernel), (ii) those of an independent numpy restatement of
the nibble stream over the ORACLE's runs (seqToRLE of the MTF index stream, RLE/Internal.hs:104-153), and
(iii) a container that decodes to the input."""
import ctypes as C
import struct

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


def nibble_stream_ref(counts, vals):
    """include/textcomp.h, sigma <= 6: dense nibble stream + escape list, as bytes"""
    c = np.asarray(counts, dtype=np.int64)
    v = np.asarray(vals, dtype=np.int64)
    first = v + np.where((c >= 2) & (c <= 4), 6, 0)
    two = (c >= 3) | (c == 0)
    second = np.where(c == 3, 12, np.where(c == 4, 13, 14))
    ln = 1 + two.astype(np.int64)
    pos = np.cumsum(ln) - ln
    total = int(ln.sum())
    padded = (total + 31) // 32 * 32
    nib = np.full(padded, 15, dtype=np.uint8)
    nib[pos] = first
    nib[pos[two] + 1] = second[two]
    body = (nib[0::2] | (nib[1::2] << 4)).astype(np.uint8).tobytes()
    esc = c[(c == 0) | (c >= 5)].astype("<u4").tobytes()
    return body + esc


def _oracle_block(text):
    L = O.bwt_encode_arr(text)
    idx, fl = O.mtf_encode_arr(L)
    counts, vals = O.rle_encode_u32_arr(idx)
    return int(np.nonzero(L < 0)[0][0]), fl, counts, vals


def _both_ways(ctx, text):
    import torch
    from textcomp import Block
    lib = ctx.lib
    n = len(text)
    d_text = torch.from_numpy(np.ascontiguousarray(text)).cuda() if n else torch.zeros(16, dtype=torch.uint8, device="cuda")
    bound = int(lib.tc_container_bound(n + 2, 257))
    a = torch.full((bound + 64,), 0xAB, dtype=torch.uint8, device="cuda")     # (dirty buffers: the call must not rely on zeros)
    b = torch.full((bound + 64,), 0xCD, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()      # the library works on its own stream: torch's fills must have landed first
    used_a = ctx.encode_container_dev(d_text.data_ptr(), n, a.data_ptr(), bound)
    cap = n + 2
    d_c = torch.empty(cap, dtype=torch.int32, device="cuda")
    d_v = torch.empty(cap, dtype=torch.int16, device="cuda")
    blk = Block()
    blk.nruns, blk.run_count, blk.run_value = cap, d_c.data_ptr(), d_v.data_ptr()
    assert lib.tc_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk)) == 0, lib.tc_last_error(ctx.handle)
    used_b = C.c_uint64(bound)
    assert lib.tc_block_to_container_dev(ctx.handle, C.byref(blk), C.c_void_p(b.data_ptr()), C.byref(used_b)) == 0
    return a[:used_a].cpu().numpy().tobytes(), b[:used_b.value].cpu().numpy().tobytes(), a, used_a


def _texts():
    r = np.random.default_rng(77)
    acgtn = np.frombuffer(b"ACGNT", np.uint8)
    out = {}
    for n in (1, 2, 15, 16, 17, 31, 33, 8191, 8192, 8193, 32767, 32768, 32769, 65536 + 5, 3 * 32768, 300001, (1 << 20) + 7):
        out["acgtn_n%d" % n] = O.gen_acgtn(0x77 + n, n)
    out["unary_100k"] = np.full(100000, 65, np.uint8)
    out["unary_tile"] = np.full(32768, 67, np.uint8)
    out["two_letters_runs"] = np.repeat(acgtn[r.integers(0, 2, 9000)], r.integers(1, 40, 9000)).astype(np.uint8)
    out["long_runs_then_noise"] = np.concatenate([np.full(70000, 71, np.uint8), O.gen_acgtn(5, 50000), np.full(40000, 84, np.uint8)])
    out["runs_of_5"] = np.repeat(acgtn[r.integers(0, 5, 40000)], 5).astype(np.uint8)
    out["runs_1_to_6"] = np.repeat(acgtn[r.integers(0, 5, 60000)], r.integers(1, 7, 60000)).astype(np.uint8)
    out["periodic_acgt"] = np.tile(np.frombuffer(b"ACGT", np.uint8), 50000)
    out["ascii"] = O.gen_ascii(9, 120000)             # sigma = 96: the two-step way inside the same call
    out["binary"] = r.integers(0, 256, 90000).astype(np.uint8)
    out["sigma7"] = np.frombuffer(b"ABCDEF", np.uint8)[r.integers(0, 6, 50000)]   # 6 letters + sentinel: byte format
    return out


TEXTS = _texts()


@pytest.mark.parametrize("name", list(TEXTS), ids=list(TEXTS))
def test_fused_container_is_the_two_step_container_and_the_oracles(ctx, name):
    text = TEXTS[name]
    fused, two, d_a, used = _both_ways(ctx, text)
    assert fused == two
    n = len(text)
    primary, fl, counts, vals = _oracle_block(text)
    magic, hn, hprim, hruns, hesc, hbody, hsum, hsigma, hfmt = struct.unpack_from("<8s6Q2I", fused, 0)
    assert magic == b"TCBLK01\0" and hn == n and hprim == primary and hruns == len(counts) and hsigma == len(fl)
    assert list(struct.unpack_from("<%dh" % hsigma, fused, 64)) == [int(x) for x in fl]
    assert hbody == len(fused) - 640
    if hsigma <= 6:
        assert hfmt == 0
        ref = nibble_stream_ref(counts, vals)
        assert hesc == int(((counts == 0) | (counts >= 5)).sum())
        assert fused[640:] == ref
    # and it decodes (device container -> block -> text)
    import torch
    from textcomp import Block
    k = int(hruns)
    o_c = torch.zeros(k + 1, dtype=torch.int32, device="cuda")
    o_v = torch.zeros(k + 1, dtype=torch.int16, device="cuda")
    d_out = torch.zeros(max(n, 1), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    out = Block()
    out.nruns, out.run_count, out.run_value = k, o_c.data_ptr(), o_v.data_ptr()
    assert ctx.lib.tc_container_to_block_dev(ctx.handle, C.c_void_p(d_a.data_ptr()), used, C.byref(out)) == 0, ctx.lib.tc_last_error(ctx.handle)
    assert np.array_equal(o_c[:k].cpu().numpy().astype(np.int64), counts)
    assert ctx.lib.tc_decode_dev(ctx.handle, C.byref(out), C.c_void_p(d_out.data_ptr())) == 0
    assert d_out[:n].cpu().numpy().tobytes() == text.tobytes()


def test_fused_container_empty_and_capacity(ctx):
    import torch
    buf = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    assert ctx.encode_container_dev(0, 0, buf.data_ptr(), 4096) == 640
    assert buf[:7].cpu().numpy().tobytes() == b"TCBLK01"
    t = torch.from_numpy(O.gen_acgtn(1, 200000)).cuda()
    used = C.c_uint64(640 + 1000)         # far too small: TC_ERR_CAPACITY and the bytes needed
    small = torch.zeros(640 + 1024, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    rc = ctx.lib.tc_encode_container_dev(ctx.handle, C.c_void_p(t.data_ptr()), 200000, C.c_void_p(small.data_ptr()), C.byref(used))
    assert rc == -2 and used.value > 640 + 60000
    need = used.value
    big = torch.zeros(need + 64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    assert ctx.encode_container_dev(t.data_ptr(), 200000, big.data_ptr(), need) == need
    used = C.c_uint64(100)
    assert ctx.lib.tc_encode_container_dev(ctx.handle, C.c_void_p(t.data_ptr()), 200000, C.c_void_p(small.data_ptr()), C.byref(used)) == -2
    assert ctx.lib.tc_encode_container_dev(ctx.handle, C.c_void_p(t.data_ptr()), 200000, C.c_void_p(small.data_ptr() + 4), C.byref(used)) == -1


@pytest.mark.parametrize("n", [1 << 24, 1 << 28])
def test_fused_container_checksum_at_scale(ctx, n):
    """both ways at 2^24 and 2^28 (the 1 GiB record: tests/test_gpu_fullsize.py): same size, same bytes"""
    import torch
    from textcomp import Block
    lib = ctx.lib
    d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert lib.tc_generate_dev(ctx.handle, 0, 0xC2, n, C.c_void_p(d_text.data_ptr())) == 0
    pcap = n + n // 4 + 4096
    a = torch.empty(pcap, dtype=torch.uint8, device="cuda")
    used_a = ctx.encode_container_dev(d_text.data_ptr(), n, a.data_ptr(), pcap)
    cap = n + 2
    d_c = torch.empty(cap, dtype=torch.int32, device="cuda")
    d_v = torch.empty(cap, dtype=torch.int16, device="cuda")
    blk = Block()
    blk.nruns, blk.run_count, blk.run_value = cap, d_c.data_ptr(), d_v.data_ptr()
    assert lib.tc_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk)) == 0
    b = torch.empty(pcap, dtype=torch.uint8, device="cuda")
    used_b = C.c_uint64(pcap)
    assert lib.tc_block_to_container_dev(ctx.handle, C.byref(blk), C.c_void_p(b.data_ptr()), C.byref(used_b)) == 0
    assert used_a == used_b.value and 0.40 * n < used_a < 0.46 * n
    assert torch.equal(a[:used_a], b[:used_a])
