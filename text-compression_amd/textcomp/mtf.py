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
    """textToBWTToMTFB (MTF.hs:96-98)."""
    return bytestringToBWTToMTFB(text.encode("utf-8"), ctx)


def bytestringToMTFB(seq, ctx=None):
    """bytestringToMTFB (MTF.hs:157-161) for single-byte elements."""
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
    """textFromBWTFromMTFB (MTF.hs:198-200)."""
    return bytestringFromBWTFromMTFB(mtf, ctx).decode("utf-8")
