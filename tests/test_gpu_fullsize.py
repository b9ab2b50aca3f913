"""BASELINE configs[2] at full size (1 GiB ACGTN record, device resident) and the largest record
the ABI admits (TC_MAX_N = 2^31 - 16 bytes, ~130 GB of workspace).

Bit-exactness against the oracle at 1 GiB: the oracle encoded this very record once
(tests/long/parity_digest.py, CPU, minutes) and its digest -- primary, sigma, final MTF list, run
count, and position-dependent 64-bit checksums of the last column, run_count[] and run_value[] --
is committed as tests/golden/c3_digest.json; the device must reproduce it.  Beyond the oracle's
reach (2^31 - 16) parity is checked through size-independent properties: the decode of the encode
is the input, bit for bit; the run lengths sum to N; exactly one primary row."""
import ctypes as C
import json
import os

import pytest

pytestmark = pytest.mark.gpu

DIGESTS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c3_digest.json")))


def _checksum(lib, ctx, tensor, nbytes):
    out = C.c_uint64()
    lib.tc_dbg_checksum64_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    rc = lib.tc_dbg_checksum64_dev(ctx.handle, C.c_void_p(tensor.data_ptr()), nbytes, C.byref(out))
    assert rc == 0, lib.tc_last_error(ctx.handle)
    return "%016x" % out.value


def _assert_digest(lib, ctx, d, blk, d_cnt, d_val, d_text):
    """the device block equals the ORACLE's block of the same record (digest of every output)"""
    import torch
    n, k = int(blk.n), int(blk.nruns)
    assert d["n"] == n
    assert int(blk.primary) == d["primary"] and int(blk.sigma) == d["sigma"] and k == d["nruns"]
    assert [int(blk.final_list[i]) for i in range(d["sigma"])] == d["final_list"]
    assert _checksum(lib, ctx, d_cnt, 4 * k) == d["run_count_checksum64"]
    if k & 1:
        d_val[k] = 0                      # the digest pads run_value[] to a whole 32-bit word
    assert _checksum(lib, ctx, d_val, 2 * (k + (k & 1))) == d["run_value_checksum64"]
    assert int(d_cnt[:k].max().item()) == d["max_run"]
    # the BWT stage on its own (tc_bwt_encode_dev): last column with byte 0 in the primary slot
    N = n + 1
    d_L = torch.zeros((N + 3) // 4 * 4, dtype=torch.uint8, device="cuda")
    prim = C.c_uint64()
    rc = lib.tc_bwt_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.c_void_p(d_L.data_ptr()), C.byref(prim))
    assert rc == 0, lib.tc_last_error(ctx.handle)
    assert prim.value == d["primary"] and int(d_L[prim.value].item()) == 0
    assert _checksum(lib, ctx, d_L, d_L.numel()) == d["last_column_checksum64"]


@pytest.mark.parametrize("n", [1 << 20, 1 << 24])
def test_digest_matches_oracle_small(n):
    """same digest machinery at sizes where tests/test_gpu_encode.py also compares element-wise"""
    import torch
    import textcomp
    from textcomp import Block
    d = DIGESTS["n%d" % n]
    ctx = textcomp.Context(0)
    lib = ctx.lib
    d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert lib.tc_generate_dev(ctx.handle, 0, d["seed"], n, C.c_void_p(d_text.data_ptr())) == 0
    cap = n + 2
    d_cnt = torch.empty(cap, dtype=torch.int32, device="cuda")
    d_val = torch.empty(cap, dtype=torch.int16, device="cuda")
    blk = Block()
    blk.nruns, blk.run_count, blk.run_value = cap, d_cnt.data_ptr(), d_val.data_ptr()
    assert lib.tc_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk)) == 0
    _assert_digest(lib, ctx, d, blk, d_cnt, d_val, d_text)
    ctx.close()


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
    del d_out
    if n == (1 << 30):      # bit-exact against the oracle's encode of this record
        _assert_digest(lib, ctx, DIGESTS["n%d" % n], blk, d_cnt, d_val, d_text)
    ctx.close()


def test_fused_container_1gib_equals_two_step():
    """the 1 GiB benchmark record encoded straight into its container (tc_encode_container_dev: the RLE stage
    writes the nibble stream, device kernels seal the container) against tc_encode_dev + tc_block_to_container_dev:
    same size, same bytes -- and the header carries the oracle's digest values for this record.
    (Other records at 2^26 .. 2^28 against the oracle: tests/test_gpu_classes_digest.py.)"""
    import struct
    import torch
    import textcomp
    from textcomp import Block
    n = 1 << 30
    d = DIGESTS["n%d" % n]
    ctx = textcomp.Context(0)
    lib = ctx.lib
    d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert lib.tc_generate_dev(ctx.handle, 0, d["seed"], n, C.c_void_p(d_text.data_ptr())) == 0
    pcap = n + n // 4 + 4096
    a = torch.empty(pcap, dtype=torch.uint8, device="cuda")
    used_a = ctx.encode_container_dev(d_text.data_ptr(), n, a.data_ptr(), pcap)
    hdr = a[:640].cpu().numpy().tobytes()
    magic, hn, hprim, hruns, hesc, hbody, hsum, hsigma, hfmt = struct.unpack_from("<8s6Q2I", hdr, 0)
    assert (hn, hprim, hruns, hsigma, hfmt) == (n, d["primary"], d["nruns"], d["sigma"], 0)
    assert list(struct.unpack_from("<%dh" % hsigma, hdr, 64)) == d["final_list"]
    cap = n + 2
    d_c = torch.empty(cap, dtype=torch.int32, device="cuda")
    d_v = torch.empty(cap, dtype=torch.int16, device="cuda")
    blk = Block()
    blk.nruns, blk.run_count, blk.run_value = cap, d_c.data_ptr(), d_v.data_ptr()
    assert lib.tc_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk)) == 0
    b = torch.empty(pcap, dtype=torch.uint8, device="cuda")
    used_b = C.c_uint64(pcap)
    assert lib.tc_block_to_container_dev(ctx.handle, C.byref(blk), C.c_void_p(b.data_ptr()), C.byref(used_b)) == 0
    assert used_a == used_b.value
    assert torch.equal(a[:used_a], b[:used_a])
    # and the container read back: the oracle's run arrays, digest for digest
    blk2 = Block()
    blk2.nruns, blk2.run_count, blk2.run_value = cap, d_c.data_ptr(), d_v.data_ptr()
    assert lib.tc_container_to_block_dev(ctx.handle, C.c_void_p(a.data_ptr()), used_a, C.byref(blk2)) == 0, lib.tc_last_error(ctx.handle)
    k = int(blk2.nruns)
    assert _checksum(lib, ctx, d_c, 4 * k) == d["run_count_checksum64"]
    if k & 1:
        d_v[k] = 0
    assert _checksum(lib, ctx, d_v, 2 * (k + (k & 1))) == d["run_value_checksum64"]
    ctx.close()
