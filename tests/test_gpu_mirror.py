"""The reference's own HUnit cases (RLE.hs:313-320, MTF.hs:287-299), run through the
host-side mirror of Data.BWT / Data.MTF / Data.RLE on the HIP library -- written to
read like the reference's tests -- plus mirror-vs-oracle checks on the reference's
value shapes."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


def _b(g):
    return [None if e is None else e.encode() for e in g]


def test_hunit_rle(golden):
    from textcomp import rle
    for v in golden["rle"]:
        assert rle.textToBWTToRLEB(v["text"]) == _b(v["rle"])          # "test 1", "test 2"
        assert rle.textFromBWTFromRLEB(_b(v["rle"])) == v["text"]      # "test 3", "test 4"


def test_hunit_mtf(golden):
    from textcomp import mtf
    v = golden["mtf"][0]
    assert mtf.textToBWTToMTFB(v["text"]) == (v["indices"], _b(v["final_list"]))   # "test 1"
    assert mtf.textFromBWTFromMTFB((v["indices"], _b(v["final_list"]))) == v["text"]  # "test 2"


def test_mirror_matches_oracle_shapes():
    from textcomp import bwt, mtf, rle
    rng = np.random.default_rng(2)
    for t in [b"", b"a", b"ba", b"abracadabra", bytes(rng.integers(0, 256, 700, dtype=np.uint8))]:
        B = bwt.bytestringToBWT(t)
        assert B == O.bytestringToBWT(t)
        assert bwt.bytestringFromWord8BWT(B) == t
        assert mtf.bytestringBWTToMTFB(B) == O.bytestringBWTToMTFB(B)
        assert mtf.bytestringToBWTToMTFB(t) == O.bytestringToBWTToMTFB(t)
        assert mtf.bytestringBWTFromMTFB(mtf.bytestringBWTToMTFB(B)) == B
        assert mtf.bytestringFromBWTFromMTFB(mtf.bytestringToBWTToMTFB(t)) == t
        R = rle.bytestringBWTToRLEB(B)
        assert R == O.bytestringBWTToRLEB(B) == rle.bytestringToBWTToRLEB(t)
        assert rle.bytestringBWTFromRLEB(R) == O.bytestringBWTFromRLEB(R)
        # Q6 class ("a", "ba": the text is its own greatest suffix): the reference's own
        # round trip is broken -- "ba" comes back as "a", "a" dies in fromJust -- and so is ours
        try:
            exp = O.bytestringFromBWTFromRLEB(R)
        except O.OracleMalformed:
            import textcomp
            with pytest.raises(textcomp.TcMalformed):
                rle.bytestringFromBWTFromRLEB(R)
        else:
            assert rle.bytestringFromBWTFromRLEB(R) == exp


def test_mirror_error_behaviour():
    import textcomp
    from textcomp import mtf, rle
    with pytest.raises(textcomp.TcMalformed):
        rle.bytestringBWTFromRLEB([b"x", b"a"])            # Prelude.read: no parse
    with pytest.raises(textcomp.TcMalformed):
        rle.bytestringBWTFromRLEB([None, b"a"])            # fromJust Nothing
    assert rle.bytestringBWTFromRLEB([b"2", b"a", b"9"]) == [97, 97]   # odd tail ignored
    with pytest.raises(textcomp.TcMalformed):
        mtf.bytestringBWTFromMTFB(([5], [b"a", None]))     # DS.index out of range


def test_hunit_through_text_variants(golden):
    """The same HUnit vectors through the Text instantiations (`...T`): the elements are the
    golden strings themselves (the reference builds them with decodeUtf8 . BS.singleton)."""
    from textcomp import mtf, rle
    for v in golden["rle"]:
        assert rle.textToBWTToRLET(v["text"]) == v["rle"]
        assert rle.bytestringToBWTToRLET(v["text"].encode()) == v["rle"]
        assert rle.textFromBWTFromRLET(v["rle"]) == v["text"]
        assert rle.bytestringFromBWTFromRLET(v["rle"]) == v["text"].encode()
    v = golden["mtf"][0]
    assert mtf.textToBWTToMTFT(v["text"]) == (v["indices"], v["final_list"])
    assert mtf.bytestringToBWTToMTFT(v["text"].encode()) == (v["indices"], v["final_list"])
    assert mtf.textFromBWTFromMTFT((v["indices"], v["final_list"])) == v["text"]
    assert mtf.bytestringFromBWTFromMTFT((v["indices"], v["final_list"])) == v["text"].encode()


def test_text_and_bytestring_variants_agree():
    """Every variant of one function family is the ByteString one re-wrapped (RLE.hs:83-274,
    MTF.hs:82-278, FMIndex.hs:385-599); non-ASCII bytes raise where decodeUtf8 would."""
    from textcomp import bwt, fmindex, mtf, rle
    text = "mississippi river banks, mississippi mud"
    bs = text.encode()
    B = bwt.textToBWT(text)                               # TextBWT = BWT Word8 of the UTF-8 bytes
    Bb = [None if v is None else bytes([v]) for v in B]   # Seq (Maybe ByteString)
    Bt = [None if v is None else chr(v) for v in B]       # Seq (Maybe Text)
    RB, RT = rle.bytestringBWTToRLEB(B), rle.bytestringBWTToRLET(B)
    assert RT == [None if e is None else e.decode() for e in RB]
    assert rle.textBWTToRLEB(B) == RB and rle.textBWTToRLET(B) == RT
    assert rle.bytestringToRLEB(Bb) == RB and rle.textToRLEB(Bt) == RB
    assert rle.bytestringToRLET(Bb) == RT and rle.textToRLET(Bt) == RT
    assert rle.bytestringBWTFromRLEB(RB) == B == rle.bytestringBWTFromRLET(RT)
    assert rle.textBWTFromRLEB(RB) == Bt == rle.textBWTFromRLET(RT)
    assert rle.bytestringFromRLEB(RB) == Bb == rle.bytestringFromRLET(RT)
    assert rle.textFromRLEB(RB) == Bt == rle.textFromRLET(RT)
    assert rle.textFromBWTFromRLET(RT) == text and rle.bytestringFromBWTFromRLET(RT) == bs
    MB, MT = mtf.bytestringBWTToMTFB(B), mtf.bytestringBWTToMTFT(B)
    assert MT == (MB[0], [None if e is None else e.decode() for e in MB[1]])
    assert mtf.textBWTToMTFB(B) == MB and mtf.textBWTToMTFT(B) == MT
    assert mtf.bytestringToMTFB(Bb) == MB and mtf.textToMTFB(Bt) == MB
    assert mtf.bytestringToMTFT(Bb) == MT and mtf.textToMTFT(Bt) == MT
    assert mtf.bytestringBWTFromMTFB(MB) == B == mtf.bytestringBWTFromMTFT(MT)
    assert mtf.textBWTFromMTFB(MB) == Bt == mtf.textBWTFromMTFT(MT)
    assert mtf.bytestringFromMTFB(MB) == Bb == mtf.bytestringFromMTFT(MT)
    assert mtf.textFromMTFB(MB) == Bt == mtf.textFromMTFT(MT)
    assert mtf.textFromBWTFromMTFT(MT) == text and mtf.bytestringFromBWTFromMTFT(MT) == bs
    pats = ["ssi", "mississippi", "zz", "i"]
    cb = fmindex.bytestringFMIndexCountS([p.encode() for p in pats], bs)
    assert fmindex.textFMIndexCountS(pats, text) == [(p, c) for p, (_, c) in zip(pats, cb)] == fmindex.textFMIndexCountP(pats, text)
    assert [c for _, c in cb] == [4, 2, None, 9]
    lb = fmindex.bytestringFMIndexLocateS([p.encode() for p in pats], bs)
    assert fmindex.textFMIndexLocateS(pats, text) == [(p, h) for p, (_, h) in zip(pats, lb)] == fmindex.textFMIndexLocateP(pats, text)
    assert fmindex.textFMIndexCountS([], text) == [] and fmindex.textFMIndexCountS(pats, "") == []
    with pytest.raises(UnicodeDecodeError):               # decodeUtf8 . BS.singleton on a byte >= 0x80
        rle.textToBWTToRLET("café")
    with pytest.raises(UnicodeDecodeError):
        fmindex.textFMIndexCountS(["a"], "café")


def test_fmindex_value_against_doc_tables(golden):
    """The FMIndex value (Cc, OccCK, SA) laid out from the device's BWT / C array / suffix array
    equals the reference's abracadabra doc tables (FMIndex/Internal.hs:49-113) and inverts."""
    from textcomp import fmindex
    d = golden["fmindex_doc"]
    text = d["text"].encode()
    cc, occck, sa = fmindex.bytestringToBWTToFMIndexB(text)
    name = lambda e: "$" if e is None else e.decode()
    assert {name(s): c for c, s in cc} == d["C"]
    assert [name(s) for _, s in cc] == sorted(d["C"], key=lambda k: (k != "$", k))      # Nothing first, then sorted
    assert "".join(name(x) for _, _, x in occck[0][1]) == d["L"]
    for sym, col in occck:
        assert [k for k, _, _ in col] == list(range(1, len(text) + 2))
        assert [o for _, o, _ in col] == d["Occ"][name(sym)]
    assert [r for r, _, _ in sa] == list(range(1, len(text) + 2))
    assert sorted(p for _, p, _ in sa) == list(range(1, len(text) + 2))
    assert all(sfx == text[p - 1:] for _, p, sfx in sa) and [s for _, _, s in sa] == sorted(s for _, _, s in sa)
    assert fmindex.bytestringFromBWTFromFMIndexB((cc, occck, sa)) == text
    assert fmindex.textToBWTToFMIndexB(d["text"]) == (cc, occck, sa)
    assert fmindex.bytestringToBWTToFMIndexB(b"") == ([], [], [])
