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
