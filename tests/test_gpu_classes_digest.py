"""Records AWAY from iid ACGTN at 2^26 .. 2^28 bytes, bit-exact against the ORACLE: the paths a non-ACGTN text takes
(timestamp MTF, dense ranks by regions, the big finish instance with its whole-bucket kernel, finish_fix tied groups,
the MSD -> LSD hand-over, sigma = 257) compared digest for digest with the oracle's encode of the same bytes
(tests/golden/classes_digest.json, written by tests/long/classes_digest.py on the CPU; the records are regenerated
here by tests/classgen.py: numpy, integer arithmetic only).  src/Data/BWT.hs:68-70 accepts any ByteString."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import classgen

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
DIGESTS = json.load(open(os.path.join(HERE, "golden", "classes_digest.json")))
KEYS = sorted((k for k in DIGESTS if not k.startswith("_")), key=lambda k: (DIGESTS[k]["n"], k))


@pytest.fixture(scope="module")
def ctx():
    import textcomp
    c = textcomp.Context(0)
    yield c
    c.close()


def _checksum(lib, ctx, tensor, nbytes):
    out = C.c_uint64()
    lib.tc_dbg_checksum64_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    rc = lib.tc_dbg_checksum64_dev(ctx.handle, C.c_void_p(tensor.data_ptr()), nbytes, C.byref(out))
    assert rc == 0, lib.tc_last_error(ctx.handle)
    return "%016x" % out.value


@pytest.mark.parametrize("key", KEYS)
def test_class_digest_equals_oracle(ctx, key):
    import torch
    from textcomp import Block
    d = DIGESTS[key]
    n = d["n"]
    lib = ctx.lib
    if d["class"] in classgen.DEV_KINDS and n >= (1 << 28):
        # (a long record of a device generator's class comes from the device generator itself: tests/test_gpu_generators.py
        # holds it equal, byte for byte, to the numpy restatement the oracle encoded -- a minute of host time saved)
        kind, seed = classgen.DEV_KINDS[d["class"]]
        d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
        assert lib.tc_generate_dev(ctx.handle, kind, seed, n, C.c_void_p(d_text.data_ptr())) == 0
    else:
        text = classgen.make(d["class"], n)
        d_text = torch.from_numpy(text).cuda()
        del text
    cap = n + 2
    d_cnt = torch.empty(cap, dtype=torch.int32, device="cuda")
    d_val = torch.empty(cap, dtype=torch.int16, device="cuda")
    blk = Block()
    blk.nruns, blk.run_count, blk.run_value = cap, d_cnt.data_ptr(), d_val.data_ptr()
    rc = lib.tc_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk))
    assert rc == 0, lib.tc_last_error(ctx.handle)
    k = int(blk.nruns)
    assert (int(blk.primary), int(blk.sigma), k) == (d["primary"], d["sigma"], d["nruns"])
    assert [int(blk.final_list[i]) for i in range(d["sigma"])] == d["final_list"]
    assert _checksum(lib, ctx, d_cnt, 4 * k) == d["run_count_checksum64"]
    if k & 1:
        d_val[k] = 0
    assert _checksum(lib, ctx, d_val, 2 * (k + (k & 1))) == d["run_value_checksum64"]
    assert int(d_cnt[:k].max().item()) == d["max_run"]
    st = ctx.stats()
    # the BWT stage on its own: the last column, byte 0 in the primary slot
    N = n + 1
    d_L = torch.zeros((N + 3) // 4 * 4, dtype=torch.uint8, device="cuda")
    prim = C.c_uint64()
    assert lib.tc_bwt_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.c_void_p(d_L.data_ptr()), C.byref(prim)) == 0
    assert prim.value == d["primary"]
    assert _checksum(lib, ctx, d_L, d_L.numel()) == d["last_column_checksum64"]
    # which way the record took (so that a change of the selectors cannot silently move it off the path it pins)
    if n >= (1 << 26):
        if d["class"] == "acgt4":
            assert st.msd_path == 1
        if d["class"] in ("zipf_words", "genome_like"):
            assert st.msd_path == 0 and st.rounds >= 2
        if d["class"] == "dev_gaps":
            # round 4: the runs of 'N' are one over-long bucket -> LSD way, sparse ranks; their members shed next to nothing
            # per plain round -> a chain round with SPARSE ranks (flags from the members through rank_of)
            assert st.chain_rounds >= 1 and st.rounds <= 8, (st.chain_rounds, st.rounds)
        if d["class"] == "dev_periodic":
            # round 4: the first doubling round sheds next to nothing -> the second one is a CHAIN round (tc_chain.hpp) and
            # the record is done in 3-4 rounds instead of log2(n / 21) + 2; the same record with the chain rounds off, same digest
            assert st.chain_rounds >= 1 and st.rounds <= 6, (st.chain_rounds, st.rounds)
            os.environ["TC_SA_CHAIN"] = "0"
            try:
                blk2 = Block()
                blk2.nruns, blk2.run_count, blk2.run_value = cap, d_cnt.data_ptr(), d_val.data_ptr()
                assert lib.tc_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk2)) == 0, lib.tc_last_error(ctx.handle)
            finally:
                os.environ.pop("TC_SA_CHAIN", None)
            st2 = ctx.stats()
            assert st2.chain_rounds == 0 and st2.rounds > 10
            assert (int(blk2.primary), int(blk2.nruns)) == (d["primary"], d["nruns"])
            assert _checksum(lib, ctx, d_cnt, 4 * k) == d["run_count_checksum64"]
        if d["class"] == "genome_like":
            # round 4: the same record by the MSD way with the big finish instance forced -- its whole buckets (the repeat
            # family, poly-A) go through the KEY ROUND (ordered by the key's remaining 32 bits before any rank exists) --
            # against the same oracle digest
            os.environ.update({"TC_SA_MSD": "2", "TC_SA_MSD_BIG": "1"})
            try:
                blk2 = Block()
                blk2.nruns, blk2.run_count, blk2.run_value = cap, d_cnt.data_ptr(), d_val.data_ptr()
                assert lib.tc_encode_dev(ctx.handle, C.c_void_p(d_text.data_ptr()), n, C.byref(blk2)) == 0, lib.tc_last_error(ctx.handle)
            finally:
                os.environ.pop("TC_SA_MSD", None)
                os.environ.pop("TC_SA_MSD_BIG", None)
            st2 = ctx.stats()
            assert st2.msd_path == 1 and st2.rounds >= 3
            assert (int(blk2.primary), int(blk2.nruns)) == (d["primary"], d["nruns"])
            assert _checksum(lib, ctx, d_cnt, 4 * k) == d["run_count_checksum64"]
    # and the block decodes
    d_back = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert lib.tc_decode_dev(ctx.handle, C.byref(blk), C.c_void_p(d_back.data_ptr())) == 0, lib.tc_last_error(ctx.handle)
    torch.cuda.synchronize()
    assert torch.equal(d_back, d_text)
