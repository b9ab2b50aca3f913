"""Mirror of Data.RLE (reference src/Data/RLE.hs:35-62), ByteString instantiation.

`RLE ByteString` = alternating list [Just (show count), symbol, ...] with bytes|None
elements (RLE/Internal.hs:95,128)."""
import numpy as np

from . import TcMalformed
from . import bwt as _bwt
from . import default_context


def _render(counts, syms):
    out = []
    for c, s in zip(counts, syms):
        out.append(str(int(c)).encode())
        out.append(None if s < 0 else bytes([int(s)]))
    return out


def _parse(elems):
    """Pairs of the element list -> (counts, syms); a trailing odd element is ignored
    (RLE/Internal.hs:187-189); fromJust Nothing / non-numeric count raise as the
    reference does (:172-173)."""
    counts, syms = [], []
    for k in range(0, len(elems) - 1, 2):
        y1, y2 = elems[k], elems[k + 1]
        if y1 is not None and y2 is None:
            counts.append(1)
            syms.append(-1)
            continue
        if y1 is None or y2 is None:
            raise TcMalformed(-3, "fromJust Nothing (RLE/Internal.hs:172-173)")
        try:
            c = int(y1.decode())
        except ValueError:
            raise TcMalformed(-3, "Prelude.read: no parse (RLE/Internal.hs:172)")
        counts.append(max(c, 0))  # replicateM_ of a non-positive count is a no-op
        syms.append(y2[0])
    return np.array(counts, dtype=np.uint32), np.array(syms, dtype=np.int16)


def bytestringBWTToRLEB(bwt, ctx=None):
    """bytestringBWTToRLEB :: BWT Word8 -> RLE ByteString (RLE.hs:117-123)."""
    if len(bwt) == 0:
        return []
    return _render(*(ctx or default_context()).rle_encode_sym(_bwt._split(bwt)))


def bytestringToBWTToRLEB(bs, ctx=None):
    """bytestringToBWTToRLEB (RLE.hs:83-85) = bytestringBWTToRLEB . bytestringToBWT."""
    c = ctx or default_context()
    L, primary = c.bwt_encode(bs)
    if len(L) == 0:
        return []
    return _render(*c.rle_encode(L, primary))


def textToBWTToRLEB(text, ctx=None):
    """textToBWTToRLEB (RLE.hs:95-97)."""
    return bytestringToBWTToRLEB(text.encode("utf-8"), ctx)


def bytestringToRLEB(seq, ctx=None):
    """bytestringToRLEB :: Seq (Maybe ByteString) -> RLE ByteString (RLE.hs:155-159),
    single-byte elements."""
    return bytestringBWTToRLEB([None if v is None else v[0] for v in seq], ctx)


def bytestringBWTFromRLEB(rle, ctx=None):
    """bytestringBWTFromRLEB :: RLE ByteString -> BWT ByteString (RLE.hs:237-241);
    returned as a BWT Word8 (list of int|None)."""
    if len(rle) == 0:
        return []
    counts, syms = _parse(rle)
    out = (ctx or default_context()).rle_decode(counts, syms)
    return [None if v < 0 else int(v) for v in out]


def bytestringFromBWTFromRLEB(rle, ctx=None):
    """bytestringFromBWTFromRLEB (RLE.hs:184-186)."""
    return _bwt.bytestringFromWord8BWT(bytestringBWTFromRLEB(rle, ctx), ctx)


def textFromBWTFromRLEB(rle, ctx=None):
    """textFromBWTFromRLEB (RLE.hs:199-201)."""
    return bytestringFromBWTFromRLEB(rle, ctx).decode("utf-8")


# ---- the Text instantiations and the remaining ByteString ones (RLE.hs:35-62) -------------------
# A TextBWT (BWT.hs:79-81) wraps the BWT Word8 of the UTF-8 bytes, so it is the same list here.
def textBWTToRLEB(tbwt, ctx=None):
    """textBWTToRLEB :: TextBWT -> RLE ByteString (RLE.hs:107-113)."""
    return bytestringBWTToRLEB(tbwt, ctx)


def bytestringBWTToRLET(bwt, ctx=None):
    """bytestringBWTToRLET :: BWT Word8 -> RLE Text (RLE.hs:137-143)."""
    return _bwt._elems_to_text(bytestringBWTToRLEB(bwt, ctx))


def textBWTToRLET(tbwt, ctx=None):
    """textBWTToRLET :: TextBWT -> RLE Text (RLE.hs:127-133)."""
    return bytestringBWTToRLET(tbwt, ctx)


def bytestringToBWTToRLET(bs, ctx=None):
    """bytestringToBWTToRLET (RLE.hs:89-91) = bytestringBWTToRLET . bytestringToBWT."""
    return _bwt._elems_to_text(bytestringToBWTToRLEB(bs, ctx))


def textToBWTToRLET(text, ctx=None):
    """textToBWTToRLET (RLE.hs:101-103) = textBWTToRLET . textToBWT."""
    return bytestringToBWTToRLET(text.encode("utf-8"), ctx)


def textToRLEB(seq, ctx=None):
    """textToRLEB :: Seq (Maybe Text) -> RLE ByteString (RLE.hs:146-152)."""
    return bytestringToRLEB(_bwt._elems_to_bytes(seq), ctx)


def textToRLET(seq, ctx=None):
    """textToRLET :: Seq (Maybe Text) -> RLE Text (RLE.hs:162-166)."""
    return _bwt._elems_to_text(textToRLEB(seq, ctx))


def bytestringToRLET(seq, ctx=None):
    """bytestringToRLET :: Seq (Maybe ByteString) -> RLE Text (RLE.hs:169-175)."""
    return _bwt._elems_to_text(bytestringToRLEB(seq, ctx))


def _rle_to_bytes(rle):
    """fmap (fmap DTE.encodeUtf8) on an RLE Text (counts are several digits long)"""
    return [None if e is None else e.encode("utf-8") for e in rle]


def bytestringBWTFromRLET(rle, ctx=None):
    """bytestringBWTFromRLET :: RLE Text -> BWT ByteString (RLE.hs:219-224); as a BWT Word8."""
    return bytestringBWTFromRLEB(_rle_to_bytes(rle), ctx)


def textBWTFromRLET(rle, ctx=None):
    """textBWTFromRLET :: RLE Text -> BWT Text (RLE.hs:211-215): str|None elements."""
    return _bwt._elems_to_text(_bwt._word8_to_bytes(bytestringBWTFromRLET(rle, ctx)))


def textBWTFromRLEB(rle, ctx=None):
    """textBWTFromRLEB :: RLE ByteString -> BWT Text (RLE.hs:228-233)."""
    return _bwt._elems_to_text(_bwt._word8_to_bytes(bytestringBWTFromRLEB(rle, ctx)))


def bytestringFromBWTFromRLET(rle, ctx=None):
    """bytestringFromBWTFromRLET :: RLE Text -> ByteString (RLE.hs:190-195)."""
    return _bwt.bytestringFromWord8BWT(bytestringBWTFromRLET(rle, ctx), ctx)


def textFromBWTFromRLET(rle, ctx=None):
    """textFromBWTFromRLET :: RLE Text -> Text (RLE.hs:205-207)."""
    return bytestringFromBWTFromRLET(rle, ctx).decode("utf-8")


def bytestringFromRLEB(rle, ctx=None):
    """bytestringFromRLEB :: RLE ByteString -> Seq (Maybe ByteString) (RLE.hs:254-258)."""
    return _bwt._word8_to_bytes(bytestringBWTFromRLEB(rle, ctx))


def textFromRLEB(rle, ctx=None):
    """textFromRLEB :: RLE ByteString -> Seq (Maybe Text) (RLE.hs:245-250)."""
    return _bwt._elems_to_text(bytestringFromRLEB(rle, ctx))


def bytestringFromRLET(rle, ctx=None):
    """bytestringFromRLET :: RLE Text -> Seq (Maybe ByteString) (RLE.hs:270-275)."""
    return _bwt._word8_to_bytes(bytestringBWTFromRLET(rle, ctx))


def textFromRLET(rle, ctx=None):
    """textFromRLET :: RLE Text -> Seq (Maybe Text) (RLE.hs:262-266)."""
    return _bwt._elems_to_text(bytestringFromRLET(rle, ctx))
