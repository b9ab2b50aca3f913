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


# ---- shared by the Text (`...T`) variants of Data.RLE / Data.MTF / Data.FMIndex ---------------
# The reference turns every byte of a BWT into a Text with `DTE.decodeUtf8 . BS.singleton`
# (RLE.hs:137-140, MTF.hs:137-140, FMIndex.hs:150-158): defined for ASCII bytes only, an exception
# otherwise.  Here: bytes <-> str element-wise with Python's strict UTF-8 codec (UnicodeDecodeError
# where decodeUtf8 throws).  Elements are the single bytes every BWT-derived sequence holds.
def _elems_to_text(seq):
    """fmap (fmap DTE.decodeUtf8)"""
    return [None if e is None else e.decode("utf-8") for e in seq]


def _elems_to_bytes(seq):
    """fmap (fmap DTE.encodeUtf8); elements must be one byte long once encoded"""
    out = []
    for e in seq:
        if e is None:
            out.append(None)
            continue
        b = e.encode("utf-8")
        if len(b) != 1:
            raise ValueError("only one-byte elements are supported on the device path: %r" % (e,))
        out.append(b)
    return out


def _word8_to_bytes(bwt):
    """BWT Word8 (int|None) -> BWT ByteString (bytes|None): fmap (fmap BS.singleton)"""
    return [None if v is None else bytes([v]) for v in bwt]
