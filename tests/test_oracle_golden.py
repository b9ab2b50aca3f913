"""Pins the CPU oracle against every known-answer vector the reference holds for
this path: the four RLE HUnit cases (RLE.hs:313-320), the two MTF
HUnit cases (MTF.hs:287-299) and the abracadabra doc tables
(FMIndex/Internal.hs:49-113).  CPU only."""
import numpy as np
import pytest

import oracle as O


def _rle_golden(g):
    return [None if e is None else e.encode() for e in g]


@pytest.mark.parametrize("k", [0, 1])
def test_rle_to(golden, k):
    # RLE.hs:316-317  rleK == textToBWTToRLE{T,B} sK   (ASCII: Text == ByteString bytes)
    v = golden["rle"][k]
    assert O.bytestringToBWTToRLEB(v["text"].encode()) == _rle_golden(v["rle"])


@pytest.mark.parametrize("k", [0, 1])
def test_rle_from(golden, k):
    # RLE.hs:318-319  sK == textFromBWTFromRLET rleK
    v = golden["rle"][k]
    assert O.bytestringFromBWTFromRLEB(_rle_golden(v["rle"])) == v["text"].encode()


def test_mtf_to(golden):
    # MTF.hs:290-293
    v = golden["mtf"][0]
    idx, fl = O.bytestringToBWTToMTFB(v["text"].encode())
    assert idx == v["indices"]
    assert fl == [None if e is None else e.encode() for e in v["final_list"]]


def test_mtf_from(golden):
    # MTF.hs:294-298
    v = golden["mtf"][0]
    fl = [None if e is None else e.encode() for e in v["final_list"]]
    assert O.bytestringFromBWTFromMTFB((v["indices"], fl)) == v["text"].encode()


def test_fmindex_doc_tables(golden):
    # FMIndex/Internal.hs:49-113 (not executed by the reference; documented values)
    d = golden["fmindex_doc"]
    L = O.bwt_encode_arr(d["text"].encode())
    assert "".join("$" if v < 0 else chr(v) for v in L) == d["L"]
    cs, cv = O.fm_cc(L)
    assert {("$" if s < 0 else chr(s)): int(v) for s, v in zip(cs, cv)} == d["C"]
    cs, occ = O.fm_occ(L)
    for r, s in enumerate(cs):
        assert occ[r].tolist() == d["Occ"]["$" if s < 0 else chr(s)]


def test_count_doc_example(golden):
    # wikipedia example the doc comment cites: "abracadabra"; Q10 quirks
    fm = O.FMIndex(b"abracadabra")
    assert fm.count(b"abra") == 2 and fm.count(b"a") == 5 and fm.count(b"abracadabra") == 1
    assert fm.count(b"") is None            # :348
    assert fm.count(b"x") is None           # absent on the first step
    assert fm.count(b"xra") == 2            # absent on a later step: loop stops (:421)
    assert fm.count(b"rab") is None         # zero matches => Nothing, never Just 0
    assert sorted(fm.locate(b"abra")) == [1, 8]
    assert O.bytestringFMIndexCountS([], b"abc") == [] and O.bytestringFMIndexCountS([b"a"], b"") == []
