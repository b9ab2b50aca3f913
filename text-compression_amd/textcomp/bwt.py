"""Mirror of Data.BWT (reference src/Data/BWT.hs:26-37) over the HIP library.

A `BWT Word8` is a list of int|None (None = Nothing, the sentinel slot)."""
import numpy as np

from . import default_context


def _to_seq(L, primary):
    out = [int(v) for v in L]
    if primary is not None:
        out[primary] = None
    return out


def _split(bwt):
    """Seq (Maybe Word8) -> (int16 array); -1 = Nothing."""
    return np.array([-1 if v is None else v for v in bwt], dtype=np.int16)


def bytestringToBWT(bs, ctx=None):
    """bytestringToBWT :: ByteString -> BWT Word8 (BWT.hs:68-70)."""
    L, primary = (ctx or default_context()).bwt_encode(bs)
    return _to_seq(L, primary)


def toBWT(xs, ctx=None):
    """toBWT specialised to Word8 lists (BWT.hs:55-64)."""
    return bytestringToBWT(bytes(xs), ctx)


def textToBWT(text, ctx=None):
    """textToBWT (BWT.hs:79-81): UTF-8 encode, then bytestringToBWT."""
    return bytestringToBWT(text.encode("utf-8"), ctx)


def bytestringFromWord8BWT(bwt, ctx=None):
    """bytestringFromWord8BWT :: BWT Word8 -> ByteString (BWT.hs:108-110)."""
    if len(bwt) == 0:
        return b""
    return (ctx or default_context()).bwt_decode_sym(_split(bwt))


def fromBWT(bwt, ctx=None):
    """fromBWT specialised to Word8 (BWT.hs:93-104)."""
    return list(bytestringFromWord8BWT(bwt, ctx))


def bytestringFromByteStringBWT(bwt, ctx=None):
    """bytestringFromByteStringBWT (BWT.hs:114-116) for single-byte elements."""
    return bytestringFromWord8BWT([None if v is None else v[0] for v in bwt], ctx)


def textFromBWT(bwt, ctx=None):
    """textFromBWT (BWT.hs:120-123)."""
    return bytestringFromWord8BWT(bwt, ctx).decode("utf-8")
