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
    """textFromBWTFromRLEB (RLE.hs:198-200)."""
    return bytestringFromBWTFromRLEB(rle, ctx).decode("utf-8")
