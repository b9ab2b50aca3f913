"""Mirror of Data.MTF (reference src/Data/MTF.hs:36-63), ByteString instantiation.

`MTF ByteString` = (indices :: [int], final list :: [bytes|None])."""
import numpy as np

from . import bwt as _bwt
from . import default_context


def _wrap(idx, fl):
    return [int(v) for v in idx], [None if v < 0 else bytes([int(v)]) for v in fl]


def bytestringBWTToMTFB(bwt, ctx=None):
    """bytestringBWTToMTFB :: BWT Word8 -> MTF ByteString (MTF.hs:117-122)."""
    if len(bwt) == 0:
        return [], []
    return _wrap(*(ctx or default_context()).mtf_encode_sym(_bwt._split(bwt)))


def bytestringToBWTToMTFB(bs, ctx=None):
    """bytestringToBWTToMTFB (MTF.hs:82-84) = bytestringBWTToMTFB . bytestringToBWT."""
    c = ctx or default_context()
    L, primary = c.bwt_encode(bs)
    if len(L) == 0:
        return [], []
    return _wrap(*c.mtf_encode(L, primary))


def textToBWTToMTFB(text, ctx=None):
    """textToBWTToMTFB (MTF.hs:94-96)."""
    return bytestringToBWTToMTFB(text.encode("utf-8"), ctx)


def bytestringToMTFB(seq, ctx=None):
    """bytestringToMTFB (MTF.hs:155-159) for single-byte elements."""
    return bytestringBWTToMTFB([None if v is None else v[0] for v in seq], ctx)


def bytestringBWTFromMTFB(mtf, ctx=None):
    """bytestringBWTFromMTFB :: MTF ByteString -> BWT ByteString (MTF.hs:240-245);
    returned as a BWT Word8 (list of int|None)."""
    idx, fl = mtf
    if len(idx) == 0 or len(fl) == 0:
        return []
    out = (ctx or default_context()).mtf_decode(np.asarray(idx, dtype=np.uint16),
                                                [-1 if v is None else v[0] for v in fl])
    return [None if v < 0 else int(v) for v in out]


def bytestringFromBWTFromMTFB(mtf, ctx=None):
    """bytestringFromBWTFromMTFB (MTF.hs:184-186)."""
    return _bwt.bytestringFromWord8BWT(bytestringBWTFromMTFB(mtf, ctx), ctx)


def textFromBWTFromMTFB(mtf, ctx=None):
    """textFromBWTFromMTFB (MTF.hs:199-201)."""
    return bytestringFromBWTFromMTFB(mtf, ctx).decode("utf-8")


# ---- the Text instantiations and the remaining ByteString ones (MTF.hs:36-63) -------------------
# MTF Text = (indices, final list of str|None); see bwt._elems_to_text for the byte <-> Text rule.
def _mtf_to_text(mtf):
    idx, fl = mtf
    return idx, _bwt._elems_to_text(fl)


def _mtf_to_bytes(mtf):
    idx, fl = mtf
    return idx, _bwt._elems_to_bytes(fl)


def textBWTToMTFB(tbwt, ctx=None):
    """textBWTToMTFB :: TextBWT -> MTF ByteString (MTF.hs:106-113)."""
    return bytestringBWTToMTFB(tbwt, ctx)


def bytestringBWTToMTFT(bwt, ctx=None):
    """bytestringBWTToMTFT :: BWT Word8 -> MTF Text (MTF.hs:137-143)."""
    return _mtf_to_text(bytestringBWTToMTFB(bwt, ctx))


def textBWTToMTFT(tbwt, ctx=None):
    """textBWTToMTFT :: TextBWT -> MTF Text (MTF.hs:126-133)."""
    return bytestringBWTToMTFT(tbwt, ctx)


def bytestringToBWTToMTFT(bs, ctx=None):
    """bytestringToBWTToMTFT (MTF.hs:88-90)."""
    return _mtf_to_text(bytestringToBWTToMTFB(bs, ctx))


def textToBWTToMTFT(text, ctx=None):
    """textToBWTToMTFT (MTF.hs:100-102)."""
    return bytestringToBWTToMTFT(text.encode("utf-8"), ctx)


def textToMTFB(seq, ctx=None):
    """textToMTFB :: Seq (Maybe Text) -> MTF ByteString (MTF.hs:146-152)."""
    return bytestringToMTFB(_bwt._elems_to_bytes(seq), ctx)


def textToMTFT(seq, ctx=None):
    """textToMTFT :: Seq (Maybe Text) -> MTF Text (MTF.hs:162-166)."""
    return _mtf_to_text(textToMTFB(seq, ctx))


def bytestringToMTFT(seq, ctx=None):
    """bytestringToMTFT :: Seq (Maybe ByteString) -> MTF Text (MTF.hs:169-175)."""
    return _mtf_to_text(bytestringToMTFB(seq, ctx))


def bytestringBWTFromMTFT(mtf, ctx=None):
    """bytestringBWTFromMTFT :: MTF Text -> BWT ByteString (MTF.hs:220-226); as a BWT Word8."""
    return bytestringBWTFromMTFB(_mtf_to_bytes(mtf), ctx)


def textBWTFromMTFT(mtf, ctx=None):
    """textBWTFromMTFT :: MTF Text -> BWT Text (MTF.hs:211-216)."""
    return _bwt._elems_to_text(_bwt._word8_to_bytes(bytestringBWTFromMTFT(mtf, ctx)))


def textBWTFromMTFB(mtf, ctx=None):
    """textBWTFromMTFB :: MTF ByteString -> BWT Text (MTF.hs:230-236)."""
    return _bwt._elems_to_text(_bwt._word8_to_bytes(bytestringBWTFromMTFB(mtf, ctx)))


def bytestringFromBWTFromMTFT(mtf, ctx=None):
    """bytestringFromBWTFromMTFT :: MTF Text -> ByteString (MTF.hs:190-195)."""
    return _bwt.bytestringFromWord8BWT(bytestringBWTFromMTFT(mtf, ctx), ctx)


def textFromBWTFromMTFT(mtf, ctx=None):
    """textFromBWTFromMTFT :: MTF Text -> Text (MTF.hs:205-207)."""
    return bytestringFromBWTFromMTFT(mtf, ctx).decode("utf-8")


def bytestringFromMTFB(mtf, ctx=None):
    """bytestringFromMTFB :: MTF ByteString -> Seq (Maybe ByteString) (MTF.hs:259-264)."""
    return _bwt._word8_to_bytes(bytestringBWTFromMTFB(mtf, ctx))


def textFromMTFB(mtf, ctx=None):
    """textFromMTFB :: MTF ByteString -> Seq (Maybe Text) (MTF.hs:249-255)."""
    return _bwt._elems_to_text(bytestringFromMTFB(mtf, ctx))


def bytestringFromMTFT(mtf, ctx=None):
    """bytestringFromMTFT :: MTF Text -> Seq (Maybe ByteString) (MTF.hs:277-283)."""
    return _bwt._word8_to_bytes(bytestringBWTFromMTFT(mtf, ctx))


def textFromMTFT(mtf, ctx=None):
    """textFromMTFT :: MTF Text -> Seq (Maybe Text) (MTF.hs:268-273)."""
    return _bwt._elems_to_text(bytestringFromMTFT(mtf, ctx))
