"""ctypes binding of oracle/liboracle.so -- the CPU restatement of the reference.

TEST INFRASTRUCTURE: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only.  The product package never imports this module.

Values use the reference's shapes: a `Seq (Maybe Word8)` is a list of
`int | None`; RLE output is the alternating `[count-as-decimal-bytes, symbol, ...]`
list of `bytes | None` the Haskell `RLE ByteString` holds (RLE/Internal.hs:95,128).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ODIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

ERR_MALFORMED = -3


class OracleMalformed(Exception):
    """The reference would throw here (fromJust / DS.index / read)."""


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_ODIR, "liboracle.so")
        src = os.path.join(_ODIR, "tc_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _ODIR, "liboracle.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        i64, i32, p = C.c_int64, C.c_int32, C.c_void_p
        L.orc_suffix_array_naive.argtypes = [p, i64, p]
        L.orc_suffix_array.argtypes = [p, i64, p]
        L.orc_sa_to_bwt.argtypes = [p, i64, p, p]
        L.orc_bwt_encode.argtypes = [p, i64, p]; L.orc_bwt_encode.restype = i64
        L.orc_bwt_decode.argtypes = [p, i64, p]; L.orc_bwt_decode.restype = i64
        L.orc_mtf_encode.argtypes = [p, i64, p, p]; L.orc_mtf_encode.restype = i32
        L.orc_mtf_decode.argtypes = [p, i64, p, i32, p]; L.orc_mtf_decode.restype = i64
        L.orc_rle_encode.argtypes = [p, i64, p, p]; L.orc_rle_encode.restype = i64
        L.orc_rle_decode.argtypes = [p, p, i64, p]; L.orc_rle_decode.restype = i64
        L.orc_rle_encode_u32.argtypes = [p, i64, p, p]; L.orc_rle_encode_u32.restype = i64
        L.orc_fm_cc.argtypes = [p, i64, p, p]; L.orc_fm_cc.restype = i32
        L.orc_fm_occ.argtypes = [p, i64, i32, p, p]
        L.orc_fm_build.argtypes = [p, i64]; L.orc_fm_build.restype = p
        L.orc_fm_free.argtypes = [p]
        L.orc_fm_count.argtypes = [p, p, i64]; L.orc_fm_count.restype = i64
        L.orc_fm_count_batch.argtypes = [p, p, p, i64, p]; L.orc_fm_count_batch.restype = None
        L.orc_fm_locate.argtypes = [p, p, i64, p, i64]; L.orc_fm_locate.restype = i64
        L.orc_gen_acgtn.argtypes = [C.c_uint64, i64, p]
        L.orc_gen_ascii.argtypes = [C.c_uint64, i64, p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b


# ---------------------------------------------------------------- array level
def suffix_array(text, naive=False):
    t = _u8(text)
    sa = np.empty(len(t) + 1, dtype=np.int32)
    (lib().orc_suffix_array_naive if naive else lib().orc_suffix_array)(_p(t), len(t), _p(sa))
    return sa


def bwt_encode_arr(text):
    """-> int16[N] (-1 = Nothing); empty for empty input."""
    t = _u8(text)
    L = np.empty(len(t) + 1, dtype=np.int16)
    N = lib().orc_bwt_encode(_p(t), len(t), _p(L))
    return L[:N]


def bwt_decode_arr(L):
    L = np.ascontiguousarray(L, dtype=np.int16)
    out = np.empty(max(len(L), 1), dtype=np.uint8)
    r = lib().orc_bwt_decode(_p(L), len(L), _p(out))
    if r == ERR_MALFORMED:
        raise OracleMalformed("fromJust Nothing (BWT/Internal.hs:195)")
    return out[:r].tobytes()


def mtf_encode_arr(x):
    x = np.ascontiguousarray(x, dtype=np.int16)
    idx = np.empty(len(x), dtype=np.int32)
    fl = np.empty(257, dtype=np.int16)
    s = lib().orc_mtf_encode(_p(x), len(x), _p(idx), _p(fl))
    return idx, fl[:s].copy()


def mtf_decode_arr(idx, flist):
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    fl = np.ascontiguousarray(flist, dtype=np.int16)
    out = np.empty(len(idx), dtype=np.int16)
    r = lib().orc_mtf_decode(_p(idx), len(idx), _p(fl), len(fl), _p(out))
    if r == ERR_MALFORMED:
        raise OracleMalformed("DS.index out of range (MTF/Internal.hs:192-194)")
    return out[:r]


def rle_encode_arr(x):
    x = np.ascontiguousarray(x, dtype=np.int16)
    counts = np.empty(2 * len(x) + 2, dtype=np.int64)
    syms = np.empty(2 * len(x) + 2, dtype=np.int16)
    k = lib().orc_rle_encode(_p(x), len(x), _p(counts), _p(syms))
    return counts[:k].copy(), syms[:k].copy()


def rle_decode_arr(counts, syms):
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    syms = np.ascontiguousarray(syms, dtype=np.int16)
    n = lib().orc_rle_decode(_p(counts), _p(syms), len(counts), None)
    out = np.empty(n, dtype=np.int16)
    lib().orc_rle_decode(_p(counts), _p(syms), len(counts), _p(out))
    return out


def rle_encode_u32_arr(x):
    x = np.ascontiguousarray(x, dtype=np.int32)
    counts = np.empty(len(x) + 1, dtype=np.int64)
    vals = np.empty(len(x) + 1, dtype=np.int32)
    k = lib().orc_rle_encode_u32(_p(x), len(x), _p(counts), _p(vals))
    return counts[:k].copy(), vals[:k].copy()


def gen_acgtn(seed, n):
    out = np.empty(n, dtype=np.uint8)
    lib().orc_gen_acgtn(seed, n, _p(out))
    return out


def gen_ascii(seed, n):
    out = np.empty(n, dtype=np.uint8)
    lib().orc_gen_ascii(seed, n, _p(out))
    return out


# ------------------------------------------- reference-shaped (Seq (Maybe ..))
def seq_of(arr):
    """int16 array -> [int|None]."""
    return [None if v < 0 else int(v) for v in arr]


def arr_of(seq):
    return np.array([-1 if v is None else v for v in seq], dtype=np.int16)


def bytestringToBWT(bs):
    """Data.BWT.bytestringToBWT (BWT.hs:68-70)."""
    return seq_of(bwt_encode_arr(bs))


def bytestringFromWord8BWT(bwt):
    """Data.BWT.bytestringFromWord8BWT (BWT.hs:108-110)."""
    return bwt_decode_arr(arr_of(bwt))


def render_rle(counts, syms):
    """(counts, syms) -> RLE ByteString element list: `Just (show count)`, symbol."""
    out = []
    for c, s in zip(counts, syms):
        out.append(str(int(c)).encode())
        out.append(None if s < 0 else bytes([int(s)]))
    return out


def parse_rle(elems):
    """Inverse of render_rle with seqFromRLE's tolerance (RLE/Internal.hs:166-189):
    a trailing odd element is ignored; `read` of a non-number throws."""
    counts, syms = [], []
    for k in range(0, len(elems) - 1, 2):
        y1, y2 = elems[k], elems[k + 1]
        if y1 is not None and y2 is None:
            counts.append(1); syms.append(-1)
            continue
        if y1 is None or y2 is None:
            raise OracleMalformed("fromJust Nothing (RLE/Internal.hs:172-173)")
        try:
            counts.append(int(y1.decode()))
        except ValueError:
            raise OracleMalformed("Prelude.read: no parse (RLE/Internal.hs:172)")
        syms.append(y2[0])
    return np.array(counts, dtype=np.int64), np.array(syms, dtype=np.int16)


def bytestringBWTToRLEB(bwt):
    """Data.RLE.bytestringBWTToRLEB (RLE.hs:117-123)."""
    return render_rle(*rle_encode_arr(arr_of(bwt)))


def bytestringToBWTToRLEB(bs):
    """Data.RLE.bytestringToBWTToRLEB (RLE.hs:83-85)."""
    return bytestringBWTToRLEB(bytestringToBWT(bs))


def bytestringBWTFromRLEB(rle):
    """Data.RLE.bytestringBWTFromRLEB (RLE.hs:237-241) -> [int|None]."""
    return seq_of(rle_decode_arr(*parse_rle(rle)))


def bytestringFromBWTFromRLEB(rle):
    """Data.RLE.bytestringFromBWTFromRLEB (RLE.hs:184-186)."""
    return bytestringFromWord8BWT(bytestringBWTFromRLEB(rle))


def bytestringBWTToMTFB(bwt):
    """Data.MTF.bytestringBWTToMTFB (MTF.hs:117-122) -> (indices, final list)."""
    idx, fl = mtf_encode_arr(arr_of(bwt))
    return [int(v) for v in idx], [None if v < 0 else bytes([int(v)]) for v in fl]


def bytestringToBWTToMTFB(bs):
    """Data.MTF.bytestringToBWTToMTFB (MTF.hs:82-84)."""
    return bytestringBWTToMTFB(bytestringToBWT(bs))


def bytestringBWTFromMTFB(mtf):
    """Data.MTF.bytestringBWTFromMTFB (MTF.hs:240-245) -> [int|None]."""
    idx, fl = mtf
    return seq_of(mtf_decode_arr(idx, [-1 if v is None else v[0] for v in fl]))


def bytestringFromBWTFromMTFB(mtf):
    """Data.MTF.bytestringFromBWTFromMTFB (MTF.hs:184-186)."""
    return bytestringFromWord8BWT(bytestringBWTFromMTFB(mtf))


class FMIndex:
    """bytestringToBWTToFMIndexB (FMIndex.hs:108-111) as an opaque handle."""

    def __init__(self, text):
        self._t = _u8(text).copy()
        self._h = lib().orc_fm_build(_p(self._t), len(self._t))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_fm_free(self._h)
            self._h = None

    def count(self, pat):
        """countFMIndex (FMIndex/Internal.hs:347-438): int, or None for Nothing."""
        p = _u8(pat)
        r = lib().orc_fm_count(self._h, _p(p) if len(p) else None, len(p))
        return None if r == 0 else int(r)

    def count_batch(self, flat, offs, threads=1):
        """countFMIndex over a batch (flat u8 bytes, int64 offs[npat + 1]) -> int64[npat], 0 = Nothing;
        threads > 1: contiguous slices on that many threads (the reference's parListChunk,
        FMIndex.hs:417-423; ctypes releases the GIL during the call)."""
        flat = np.ascontiguousarray(flat, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        npat = len(offs) - 1
        out = np.zeros(npat, dtype=np.int64)
        if threads <= 1:
            lib().orc_fm_count_batch(self._h, _p(flat), _p(offs), npat, _p(out))
            return out
        import threading
        per = (npat + threads - 1) // threads
        ths = []
        for t in range(threads):
            lo, hi = min(t * per, npat), min((t + 1) * per, npat)
            if hi > lo:
                ths.append(threading.Thread(target=lib().orc_fm_count_batch, args=(
                    self._h, _p(flat), C.c_void_p(offs.ctypes.data + 8 * lo), hi - lo, C.c_void_p(out.ctypes.data + 8 * lo))))
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        return out

    def locate(self, pat):
        p = _u8(pat)
        cap = len(self._t) + 1
        out = np.empty(cap, dtype=np.int64)
        k = lib().orc_fm_locate(self._h, _p(p) if len(p) else None, len(p), _p(out), cap)
        return [int(v) for v in out[:k]]


def bytestringFMIndexCountS(pats, text):
    """Data.FMIndex.bytestringFMIndexCountS (FMIndex.hs:362-379)."""
    if not pats or len(text) == 0:
        return []
    fm = FMIndex(text)
    return [(p, fm.count(p)) for p in pats]


def fm_cc(L):
    L = np.ascontiguousarray(L, dtype=np.int16)
    cs = np.empty(257, dtype=np.int16)
    cv = np.empty(257, dtype=np.int64)
    s = lib().orc_fm_cc(_p(L), len(L), _p(cs), _p(cv))
    return cs[:s].copy(), cv[:s].copy()


def fm_occ(L):
    L = np.ascontiguousarray(L, dtype=np.int16)
    cs, _ = fm_cc(L)
    occ = np.empty((len(cs), len(L)), dtype=np.int32)
    lib().orc_fm_occ(_p(L), len(L), len(cs), _p(cs), _p(occ))
    return cs, occ
