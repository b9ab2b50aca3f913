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
