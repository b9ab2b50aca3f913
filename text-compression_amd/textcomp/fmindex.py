"""Mirror of Data.FMIndex count / locate (reference src/Data/FMIndex.hs:49-82),
ByteString instantiation."""
from . import default_context


def bytestringFMIndexCountS(pats, text, ctx=None):
    """bytestringFMIndexCountS :: [ByteString] -> ByteString -> Seq (ByteString, Maybe Int)
    (FMIndex.hs:362-379).  Empty pattern list or empty input => empty result."""
    if len(pats) == 0 or len(text) == 0:
        return []
    fm = (ctx or default_context()).fm_build(text)
    try:
        counts = fm.count(pats)
    finally:
        fm.close()
    return [(p, None if c == 0 else int(c)) for p, c in zip(pats, counts)]


def bytestringFMIndexCountP(pats, text, ctx=None):
    """bytestringFMIndexCountP (FMIndex.hs:411-432): same values, same order; the
    spark pool over patterns is one batched launch here."""
    return bytestringFMIndexCountS(pats, text, ctx)


def bytestringFMIndexLocateS(pats, text, ctx=None):
    """bytestringFMIndexLocateS (FMIndex.hs:475-497): 1-based positions, SA order."""
    if len(pats) == 0 or len(text) == 0:
        return []
    fm = (ctx or default_context()).fm_build(text)
    try:
        hits = fm.locate(pats)
    finally:
        fm.close()
    return [(p, [int(v) for v in h]) for p, h in zip(pats, hits)]


def bytestringFMIndexLocateP(pats, text, ctx=None):
    """bytestringFMIndexLocateP (FMIndex.hs:538-563)."""
    return bytestringFMIndexLocateS(pats, text, ctx)


# ---- Text instantiations (FMIndex.hs:385-403,436-462,503-530,570-599) ----------------------------
# The index is built over the UTF-8 bytes, each turned into a Text by decodeUtf8 . BS.singleton
# (ASCII only, an exception otherwise); patterns are split into characters.  For ASCII input that is
# byte matching, which is what runs here; anything else raises as the reference does.
def _ascii_bytes(text):
    b = text.encode("utf-8")
    for v in b:
        if v >= 0x80:
            bytes([v]).decode("utf-8")     # raises UnicodeDecodeError like decodeUtf8
    return b


def textFMIndexCountS(pats, text, ctx=None):
    """textFMIndexCountS :: [Text] -> Text -> Seq (Text, Maybe Int) (FMIndex.hs:385-402)."""
    if len(pats) == 0 or len(text) == 0:
        return []
    res = bytestringFMIndexCountS([p.encode("utf-8") for p in pats], _ascii_bytes(text), ctx)
    return [(p, c) for p, (_, c) in zip(pats, res)]


def textFMIndexCountP(pats, text, ctx=None):
    """textFMIndexCountP (FMIndex.hs:441-462)."""
    return textFMIndexCountS(pats, text, ctx)


def textFMIndexLocateS(pats, text, ctx=None):
    """textFMIndexLocateS (FMIndex.hs:505-527)."""
    if len(pats) == 0 or len(text) == 0:
        return []
    res = bytestringFMIndexLocateS([p.encode("utf-8") for p in pats], _ascii_bytes(text), ctx)
    return [(p, h) for p, (_, h) in zip(pats, res)]


def textFMIndexLocateP(pats, text, ctx=None):
    """textFMIndexLocateP (FMIndex.hs:574-599)."""
    return textFMIndexLocateS(pats, text, ctx)


# ---- the FMIndex VALUE (FMIndex/Internal.hs:153-170) ------------------------------------------
# FMIndex (Cc, OccCK, SA): sigma pairs, sigma x N triples and N suffix records -- O(sigma N + N^2)
# elements, a shape for small inputs (the reference builds it through the O(n^2) rotation matrix).
# The device supplies what the path computes (BWT, C array, suffix array); the Seq-of-tuples shape
# is laid out here.  count / locate never materialise it: they run on the device index (`tc_fm`).
def bytestringToBWTToFMIndexB(bs, ctx=None):
    """bytestringToBWTToFMIndexB :: ByteString -> FMIndex ByteString (FMIndex.hs:108-111,162-183):
    (Cc, OccCK, SA) with Cc = [(C[c], c)], OccCK = [(c, [(k, Occ(c, k), L[k])  k = 1..N])] for the
    present symbols c in order (Nothing first, Occ inclusive of k; seqToCc / seqToOccCK,
    FMIndex/Internal.hs:195-316) and SA = [(suffixindex, suffixstartpos, suffix)], 1-based
    (createSuffixArray, BWT/Internal.hs:110-134)."""
    from . import bwt as _bwt
    if len(bs) == 0:
        return [], [], []
    c = ctx or default_context()
    B = _bwt.bytestringToBWT(bs, c)
    fm = c.fm_build(bs)
    try:
        info = fm.info()
    finally:
        fm.close()
    el = lambda s: None if s < 0 else bytes([int(s)])
    cc = [(int(v), el(s)) for s, v in zip(info["c_sym"], info["c_val"])]
    Lb = [None if v is None else bytes([v]) for v in B]
    occck = []
    for _, sym in cc:
        run, col = 0, []
        for k, x in enumerate(Lb, start=1):
            if x == sym:
                run += 1
            col.append((k, run, x))
        occck.append((sym, col))
    sa = c.suffix_array(bs)
    sarec = [(j + 1, int(p) + 1, bytes(bs[int(p):])) for j, p in enumerate(sa)]
    return cc, occck, sarec


def textToBWTToFMIndexB(text, ctx=None):
    """textToBWTToFMIndexB (FMIndex.hs:122-125)."""
    return bytestringToBWTToFMIndexB(text.encode("utf-8"), ctx)


def bytestringFromBWTFromFMIndexB(fmi, ctx=None):
    """bytestringFromBWTFromFMIndexB (FMIndex.hs:244-246): the text back from the index -- the BWT is
    the third component of the first OccCK row (seqFromFMIndex, FMIndex/Internal.hs:324-338)."""
    from . import bwt as _bwt
    cc, occck, sa = fmi
    if len(cc) == 0 or len(occck) == 0 or len(sa) == 0:
        return b""
    return _bwt.bytestringFromByteStringBWT([x for _, _, x in occck[0][1]], ctx)
