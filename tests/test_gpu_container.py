"""Encoded-block container: header + packed runs, one byte string per record.
Round trips through the host entry points and the device ones; corruption is detected."""
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


TEXTS = [b"", b"a", b"ba", b"abracadabra", b"A" * 5000, O.gen_acgtn(3, 100000).tobytes(),
         O.gen_ascii(4, 70000).tobytes(), bytes(range(256)) * 40, b"\x00" * 33 + b"\xff" * 33]


@pytest.mark.parametrize("t", TEXTS, ids=lambda t: "n%d" % len(t))
def test_container_roundtrip_host(ctx, t):
    blob = ctx.encode_container(t)
    assert blob[:7] == b"TCBLK01" and len(blob) >= 640 and len(blob) % 4 == 0
    assert ctx.decode_container(blob) == t
    if len(t) >= 1000:
        sigma = len(set(t)) + 1
        if sigma <= 6:
            assert len(blob) < 640 + len(t)          # nibble stream: well under a byte per symbol


def test_container_matches_block(ctx):
    """device entry points: block -> container -> block is the identity (runs, header fields)."""
    import torch
    from textcomp import Block
    lib = ctx.lib
    t = O.gen_acgtn(9, 300001)
    n = len(t)
    d_text = torch.from_numpy(np.frombuffer(t.tobytes(), np.uint8).copy()).cuda()
    cap = n + 2
    d_c = torch.empty(cap, dtype=torch.int32, device="cuda"); d_v = torch.empty(cap, dtype=torch.int16, device="cuda")
    blk = Block(); blk.nruns = cap; blk.run_count = d_c.data_ptr(); blk.run_value = d_v.data_ptr()
    assert lib.tc_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk)) == 0
    k = int(blk.nruns)
    bound = lib.tc_container_bound(k, blk.sigma)
    buf = torch.zeros(bound, dtype=torch.uint8, device="cuda")
    used = C.c_uint64(bound)
    assert lib.tc_block_to_container_dev(ctx.handle, C.byref(blk), C.c_void_p(buf.data_ptr()), C.byref(used)) == 0
    assert 640 < used.value <= bound
    small = C.c_uint64(used.value - 16)
    assert lib.tc_block_to_container_dev(ctx.handle, C.byref(blk), C.c_void_p(buf.data_ptr()), C.byref(small)) == -2
    assert small.value == used.value                      # bytes needed
    assert lib.tc_block_to_container_dev(ctx.handle, C.byref(blk), C.c_void_p(buf.data_ptr()), C.byref(used)) == 0
    o_c = torch.zeros(k, dtype=torch.int32, device="cuda"); o_v = torch.zeros(k, dtype=torch.int16, device="cuda")
    out = Block(); out.nruns = k - 1; out.run_count = o_c.data_ptr(); out.run_value = o_v.data_ptr()
    assert lib.tc_container_to_block_dev(ctx.handle, C.c_void_p(buf.data_ptr()), used.value, C.byref(out)) == -2
    assert int(out.nruns) == k                            # run slots needed
    assert lib.tc_container_to_block_dev(ctx.handle, C.c_void_p(buf.data_ptr()), used.value, C.byref(out)) == 0
    assert (int(out.n), int(out.primary), int(out.sigma), int(out.nruns)) == (n, int(blk.primary), int(blk.sigma), k)
    assert list(out.final_list[:out.sigma]) == list(blk.final_list[:blk.sigma])
    assert torch.equal(o_c, d_c[:k]) and torch.equal(o_v, d_v[:k])
    d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    assert lib.tc_decode_dev(ctx.handle, C.byref(out), C.c_void_p(d_out.data_ptr())) == 0
    assert torch.equal(d_out, d_text)


def test_container_detects_corruption(ctx):
    import textcomp
    t = O.gen_acgtn(5, 50000).tobytes()
    blob = bytearray(ctx.encode_container(t))
    for pos, what in ((0, "magic"), (700, "payload"), (len(blob) - 1, "payload tail"), (8, "n field")):
        bad = bytearray(blob)
        bad[pos] ^= 0x40
        with pytest.raises(textcomp.TcError) as e:
            ctx.decode_container(bytes(bad))
        assert e.value.code == -3, what
    with pytest.raises(textcomp.TcError) as e:
        ctx.decode_container(bytes(blob[:-4]))            # truncated
    assert e.value.code == -3
    with pytest.raises(textcomp.TcError) as e:
        ctx.decode_container(bytes(blob[:100]))
    assert e.value.code == -3
    with pytest.raises(textcomp.TcError) as e:
        ctx.encode_container(t, cap=1000)                 # too small: capacity error
    assert e.value.code == -2


# ------------------------------------------------------------------ chunked stream (8f-4)
@pytest.mark.parametrize("n,block", [(0, 1000), (1, 1000), (999, 1000), (1000, 1000), (1001, 1000),
                                     (250000, 65536), (300000, 100000), (1 << 20, 0), (70001, 7)])
def test_stream_roundtrip(ctx, n, block):
    """A text cut into records of `block` bytes: every record is the container of that slice on its
    own (same bytes as tc_encode_container gives), the stream decodes to the text."""
    if block == 7:
        n = 700                                        # 100 tiny records
    t = O.gen_acgtn(41, n).tobytes() if n % 2 == 0 else O.gen_ascii(42, n).tobytes()
    blob = ctx.encode_stream(t, block)
    b = block or (1 << 30)
    nb = max(1, -(-n // b))
    assert ctx.stream_info(blob) == (n, nb)
    exp = b"".join(ctx.encode_container(t[k * b:(k + 1) * b]) for k in range(nb))
    assert blob == exp
    assert ctx.decode_stream(blob) == t


def test_stream_capacity_and_corruption(ctx):
    import textcomp
    lib = ctx.lib
    t = np.frombuffer(O.gen_acgtn(43, 200000).tobytes(), np.uint8).copy()
    blob = ctx.encode_stream(t.tobytes(), 50000)
    out = np.empty(len(blob), np.uint8)
    used = C.c_uint64(len(blob) - 1)                       # one byte short: the last container does not fit
    rc = lib.tc_encode_stream(ctx.handle, C.c_void_p(t.ctypes.data), len(t), 50000, C.c_void_p(out.ctypes.data), C.byref(used))
    assert rc == -2 and used.value == lib.tc_stream_bound(len(t), 50000) >= len(blob)
    used = C.c_uint64(len(blob))
    rc = lib.tc_encode_stream(ctx.handle, C.c_void_p(t.ctypes.data), len(t), 50000, C.c_void_p(out.ctypes.data), C.byref(used))
    assert rc == 0 and out[:used.value].tobytes() == blob
    # decode: text buffer too small -> bytes needed
    b = np.frombuffer(blob, np.uint8)
    txt = np.empty(len(t), np.uint8)
    got = C.c_uint64(len(t) - 1)
    assert lib.tc_decode_stream(ctx.handle, C.c_void_p(b.ctypes.data), len(b), C.c_void_p(txt.ctypes.data), C.byref(got)) == -2
    assert got.value == len(t)
    # a flipped payload byte in the third record, a truncated stream, a broken magic
    bad = bytearray(blob)
    third = 2 * (len(blob) // 4) + 700
    bad[third] ^= 0x10
    with pytest.raises(textcomp.TcMalformed):
        ctx.decode_stream(bytes(bad))
    with pytest.raises(textcomp.TcMalformed):
        ctx.decode_stream(blob[:-4])
    with pytest.raises(textcomp.TcMalformed):
        ctx.decode_stream(b"XX" + blob[2:])
    assert lib.tc_encode_stream(ctx.handle, C.c_void_p(t.ctypes.data), len(t), 1 << 31, C.c_void_p(out.ctypes.data), C.byref(used)) == -1
