"""textcomp -- host-side mirror of the reference's Data.BWT / Data.MTF / Data.RLE /
Data.FMIndex surface over libtextcomp.so (HIP, gfx950).

Two levels:
  * `Context`: array-level calls (numpy in / numpy out) straight onto the C ABI.
  * `textcomp.bwt / .mtf / .rle / .fmindex`: functions with the reference's names
    and value shapes (`Seq (Maybe Word8)` = list of int|None, `RLE ByteString` =
    alternating [b"count", symbol] list, ...), for callers and parity tests.

There is no CPU path: without the built library and a usable GPU everything raises.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Block, Stats, TcError, TcMalformed  # noqa: F401


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u8(b):
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(bytes(b), dtype=np.uint8)


class Context:
    """One device + stream + workspace (`tc_ctx`)."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.tc_ctx_create(device, C.byref(h))
        if rc != 0:
            raise TcError(rc, "tc_ctx_create(device=%d): no usable HIP device" % device)
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.tc_last_error(self._h).decode(errors="replace")
            raise (TcMalformed if rc == _lib.TC_ERR_MALFORMED else TcError)(rc, msg)

    @property
    def handle(self):
        return self._h

    @property
    def lib(self):
        return self._lib

    def place_workspace(self, d_text_ptr, n, blk, tries=5):
        """tc_ctx_place_workspace: encode the record at device address d_text_ptr on up to `tries` workspace
        placements, keep the fastest; returns (ms per placement tried, chosen index)."""
        ms = (C.c_double * 8)()
        ch = C.c_int(-1)
        self._check(self._lib.tc_ctx_place_workspace(self._h, C.c_void_p(d_text_ptr), n, C.byref(blk), tries, ms, C.byref(ch)))
        return [x for x in ms if x > 0], ch.value

    def stats(self):
        s = Stats()
        self._check(self._lib.tc_get_stats(self._h, C.byref(s)))
        return s

    # ---------------------------------------------------------------- BWT
    def bwt_encode(self, text):
        """-> (L uint8[n+1], primary); empty input -> (empty, None)."""
        t = _u8(text)
        n = len(t)
        if n == 0:
            return np.empty(0, np.uint8), None
        L = np.empty(n + 1, np.uint8)
        prim = C.c_uint64()
        self._check(self._lib.tc_bwt_encode(self._h, _ptr(t), n, _ptr(L), C.byref(prim)))
        return L, int(prim.value)

    def suffix_array(self, text):
        t = _u8(text)
        sa = np.empty(len(t) + 1, np.uint32)
        self._check(self._lib.tc_suffix_array(self._h, _ptr(t) if len(t) else None, len(t), _ptr(sa)))
        return sa

    def bwt_decode(self, L, primary):
        L = _u8(L)
        N = len(L)
        if N == 0:
            return b""
        out = np.empty(max(N - 1, 1), np.uint8)
        self._check(self._lib.tc_bwt_decode(self._h, _ptr(L), N, primary, _ptr(out)))
        return out[:N - 1].tobytes()

    def bwt_decode_sym(self, sym):
        sym = np.ascontiguousarray(sym, dtype=np.int16)
        N = len(sym)
        if N == 0:
            return b""
        out = np.empty(N, np.uint8)
        n_out = C.c_uint64()
        self._check(self._lib.tc_bwt_decode_sym(self._h, _ptr(sym), N, _ptr(out), C.byref(n_out)))
        return out[:n_out.value].tobytes()

    # ---------------------------------------------------------------- MTF
    def mtf_encode(self, L, primary):
        """(L, primary|None) -> (idx uint16[N], final_list int16[sigma])."""
        L = _u8(L)
        N = len(L)
        if N == 0:
            return np.empty(0, np.uint16), np.empty(0, np.int16)
        idx = np.empty(N, np.uint16)
        fl = np.empty(_lib.TC_MAX_SIGMA, np.int16)
        sig = C.c_uint32()
        self._check(self._lib.tc_mtf_encode(self._h, _ptr(L), N, -1 if primary is None else primary,
                                            _ptr(idx), _ptr(fl), C.byref(sig)))
        return idx, fl[:sig.value].copy()

    def mtf_encode_sym(self, sym):
        sym = np.ascontiguousarray(sym, dtype=np.int16)
        N = len(sym)
        if N == 0:
            return np.empty(0, np.uint16), np.empty(0, np.int16)
        idx = np.empty(N, np.uint16)
        fl = np.empty(_lib.TC_MAX_SIGMA, np.int16)
        sig = C.c_uint32()
        self._check(self._lib.tc_mtf_encode_sym(self._h, _ptr(sym), N, _ptr(idx), _ptr(fl),
                                                C.byref(sig)))
        return idx, fl[:sig.value].copy()

    def mtf_decode(self, idx, flist):
        idx = np.ascontiguousarray(idx, dtype=np.uint16)
        fl = np.ascontiguousarray(flist, dtype=np.int16)
        if len(idx) == 0 or len(fl) == 0:
            return np.empty(0, np.int16)
        out = np.empty(len(idx), np.int16)
        self._check(self._lib.tc_mtf_decode(self._h, _ptr(idx), len(idx), _ptr(fl), len(fl), _ptr(out)))
        return out

    # ---------------------------------------------------------------- RLE
    def _rle_call(self, fn, args_before, N, sym_dtype, cap=None):
        cap = (2 * N + 2) if cap is None else cap
        counts = np.empty(max(cap, 1), np.uint32)
        syms = np.empty(max(cap, 1), sym_dtype)
        nr = C.c_uint64(cap)
        self._check(fn(self._h, *args_before, _ptr(counts), _ptr(syms), C.byref(nr)))
        return counts[:nr.value].copy(), syms[:nr.value].copy()

    def rle_encode(self, L, primary, cap=None):
        L = _u8(L)
        if len(L) == 0:
            return np.empty(0, np.uint32), np.empty(0, np.int16)
        return self._rle_call(self._lib.tc_rle_encode,
                              (_ptr(L), len(L), -1 if primary is None else primary), len(L),
                              np.int16, cap)

    def rle_encode_sym(self, sym, cap=None):
        sym = np.ascontiguousarray(sym, dtype=np.int16)
        if len(sym) == 0:
            return np.empty(0, np.uint32), np.empty(0, np.int16)
        return self._rle_call(self._lib.tc_rle_encode_sym, (_ptr(sym), len(sym)), len(sym), np.int16,
                              cap)

    def rle_encode_u16(self, vals, cap=None):
        vals = np.ascontiguousarray(vals, dtype=np.uint16)
        if len(vals) == 0:
            return np.empty(0, np.uint32), np.empty(0, np.uint16)
        return self._rle_call(self._lib.tc_rle_encode_u16, (_ptr(vals), len(vals)), len(vals),
                              np.uint16, cap)

    def _rle_decode(self, fn, counts, syms, dtype):
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        syms = np.ascontiguousarray(syms, dtype=dtype)
        if len(counts) == 0:
            return np.empty(0, dtype)
        cap = int(counts.astype(np.uint64).sum()) + len(counts) + 1
        out = np.empty(cap, dtype)
        N = C.c_uint64(cap)
        self._check(fn(self._h, _ptr(counts), _ptr(syms), len(counts), _ptr(out), C.byref(N)))
        return out[:N.value].copy()

    def rle_decode(self, counts, syms):
        return self._rle_decode(self._lib.tc_rle_decode, counts, syms, np.int16)

    def rle_decode_u16(self, counts, vals):
        return self._rle_decode(self._lib.tc_rle_decode_u16, counts, vals, np.uint16)

    # ------------------------------------------------------------- fused
    def encode(self, text, cap=None):
        """Fused BWT->MTF->RLE.  -> dict(n, primary, sigma, final_list, run_count, run_value)."""
        t = _u8(text)
        n = len(t)
        cap = (n + 2) if cap is None else cap
        rc_ = np.empty(max(cap, 1), np.uint32)
        rv_ = np.empty(max(cap, 1), np.uint16)
        b = Block()
        b.nruns = cap
        b.run_count = rc_.ctypes.data
        b.run_value = rv_.ctypes.data
        self._check(self._lib.tc_encode(self._h, _ptr(t) if n else None, n, C.byref(b)))
        k = int(b.nruns)
        return dict(n=int(b.n), primary=int(b.primary) if n else None, sigma=int(b.sigma),
                    final_list=np.array(b.final_list[:b.sigma], dtype=np.int16),
                    run_count=rc_[:k].copy(), run_value=rv_[:k].copy())

    def decode(self, blk):
        n = int(blk["n"])
        if n == 0:
            return b""
        rc_ = np.ascontiguousarray(blk["run_count"], dtype=np.uint32)
        rv_ = np.ascontiguousarray(blk["run_value"], dtype=np.uint16)
        b = Block()
        b.n = n
        b.primary = int(blk["primary"])
        b.sigma = int(blk["sigma"])
        for i, v in enumerate(blk["final_list"]):
            b.final_list[i] = int(v)
        b.nruns = len(rc_)
        b.run_count = rc_.ctypes.data
        b.run_value = rv_.ctypes.data
        out = np.empty(n, np.uint8)
        self._check(self._lib.tc_decode(self._h, C.byref(b), _ptr(out)))
        return out.tobytes()

    # --------------------------------------------------------- container
    def encode_container(self, text, cap=None):
        """text -> one self-describing byte string (header + packed runs); see textcomp.h."""
        t = _u8(text)
        n = len(t)
        cap = int(self._lib.tc_container_bound(n + 2, 257 if n else 0)) if cap is None else int(cap)
        out = np.empty(max(cap, 1), np.uint8)
        used = C.c_uint64(cap)
        self._check(self._lib.tc_encode_container(self._h, _ptr(t), n, _ptr(out), C.byref(used)))
        return out[:used.value].tobytes()

    def encode_container_dev(self, d_text_ptr, n, d_out_ptr, cap):
        """device text -> device container (tc_encode_container_dev: for sigma <= 6 the RLE stage writes the
        container's nibble stream itself).  Returns the bytes used."""
        used = C.c_uint64(int(cap))
        self._check(self._lib.tc_encode_container_dev(self._h, C.c_void_p(d_text_ptr), int(n), C.c_void_p(d_out_ptr),
                                                      C.byref(used)))
        return int(used.value)

    def decode_container(self, blob):
        b = np.frombuffer(bytes(blob), np.uint8)
        n, nruns = C.c_uint64(), C.c_uint64()
        self._check(self._lib.tc_container_info(self._h, _ptr(b), len(b), C.byref(n), C.byref(nruns)))
        out = np.empty(max(n.value, 1), np.uint8)
        got = C.c_uint64()
        self._check(self._lib.tc_decode_container(self._h, _ptr(b), len(b), _ptr(out), C.byref(got)))
        return out[:got.value].tobytes()

    # ------------------------------------------------------ chunked stream
    def encode_stream(self, text, block_bytes=0, cap=None):
        """text of any length -> containers of independent records of block_bytes, back to back
        (copies overlap the encode); see textcomp.h.  Without `cap` the output buffer starts at
        2 bytes per input byte and falls back to tc_stream_bound when that is too small."""
        t = _u8(text)
        n = len(t)
        bound = int(self._lib.tc_stream_bound(n, block_bytes))
        caps = [int(cap)] if cap is not None else sorted({min(bound, 2 * n + 4096 * (1 + n // max(int(block_bytes) or (1 << 30), 1))), bound})
        for i, c in enumerate(caps):
            out = np.empty(max(c, 1), np.uint8)
            used = C.c_uint64(c)
            rc = self._lib.tc_encode_stream(self._h, _ptr(t), n, int(block_bytes), _ptr(out), C.byref(used))
            if rc == _lib.TC_ERR_CAPACITY and i + 1 < len(caps):
                continue
            self._check(rc)
            return out[:used.value].tobytes()

    def stream_info(self, blob):
        """(total text bytes, number of records) of a stream."""
        b = np.frombuffer(bytes(blob), np.uint8)
        n, nb = C.c_uint64(), C.c_uint64()
        self._check(self._lib.tc_stream_info(self._h, _ptr(b), len(b), C.byref(n), C.byref(nb)))
        return n.value, nb.value

    def decode_stream(self, blob):
        b = np.frombuffer(bytes(blob), np.uint8)
        n, _ = self.stream_info(b)
        out = np.empty(max(n, 1), np.uint8)
        got = C.c_uint64(n)
        self._check(self._lib.tc_decode_stream(self._h, _ptr(b), len(b), _ptr(out), C.byref(got)))
        return out[:got.value].tobytes()

    # ----------------------------------------------------------- FM-index
    def fm_build(self, text):
        return FMIndexHandle(self, text)

    def fm_build_dev(self, d_text):
        """index of a text that already lies in HBM (a torch uint8 tensor on this context's device): tc_fm_build_dev"""
        h = C.c_void_p()
        self._check(self.lib.tc_fm_build_dev(self.handle, C.c_void_p(d_text.data_ptr()) if d_text.numel() else None, d_text.numel(), C.byref(h)))
        return FMIndexHandle(self, None, _handle=h, _n=d_text.numel())


class FMIndexHandle:
    """`tc_fm`: the device-resident FM-index of one text."""

    def __init__(self, ctx, text, _handle=None, _n=0):
        self._ctx = ctx
        if _handle is not None:      # an index that arrived from another GPU (textcomp.fmshard)
            self._h, self.n = _handle, _n
            return
        t = _u8(text)
        h = C.c_void_p()
        ctx._check(ctx.lib.tc_fm_build(ctx.handle, _ptr(t) if len(t) else None, len(t), C.byref(h)))
        self._h = h
        self.n = len(t)

    def export_dev(self, with_locate=False):
        """The index as one device byte string (torch uint8 tensor) -- what a broadcast moves."""
        import torch
        ctx = self._ctx
        nb = int(ctx.lib.tc_fm_export_bound(self._h, int(with_locate)))
        buf = torch.empty(nb, dtype=torch.uint8, device="cuda:%d" % ctx.device)
        torch.cuda.synchronize()
        used = C.c_uint64(nb)
        ctx._check(ctx.lib.tc_fm_export_dev(ctx.handle, self._h, int(with_locate), C.c_void_p(buf.data_ptr()), C.byref(used)))
        return buf[:used.value]

    @classmethod
    def import_dev(cls, ctx, buf, n=0):
        """Inverse of export_dev on this rank's device; `buf` may be released afterwards."""
        import torch
        torch.cuda.synchronize()
        h = C.c_void_p()
        ctx._check(ctx.lib.tc_fm_import_dev(ctx.handle, C.c_void_p(buf.data_ptr()), buf.numel(), C.byref(h)))
        return cls(ctx, None, _handle=h, _n=n)

    def count_dev(self, d_pats, d_offs, npat):
        """patterns resident on the device (flat uint8 tensor, uint64/int64 offsets [npat + 1]) -> int64 tensor"""
        import torch
        ctx = self._ctx
        out = torch.zeros(max(npat, 1), dtype=torch.int64, device=d_pats.device)[:npat]
        torch.cuda.synchronize()
        if npat:
            ctx._check(ctx.lib.tc_fm_count_dev(ctx.handle, self._h, C.c_void_p(d_pats.data_ptr()),
                                               C.c_void_p(d_offs.data_ptr()), npat, C.c_void_p(out.data_ptr())))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._ctx.lib.tc_fm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _pack(pats):
        offs = np.zeros(len(pats) + 1, np.uint64)
        for i, p in enumerate(pats):
            offs[i + 1] = offs[i] + len(p)
        flat = np.frombuffer(b"".join(bytes(p) for p in pats) + b"\0", dtype=np.uint8).copy()
        return flat, offs

    def count(self, pats):
        """-> int64[npat]; 0 stands for Nothing."""
        if len(pats) == 0:
            return np.empty(0, np.int64)
        flat, offs = self._pack(pats)
        out = np.empty(len(pats), np.int64)
        ctx = self._ctx
        ctx._check(ctx.lib.tc_fm_count(ctx.handle, self._h, _ptr(flat), _ptr(offs), len(pats), _ptr(out)))
        return out

    def locate(self, pats):
        """-> list of uint64 arrays (1-based positions, SA order)."""
        if len(pats) == 0:
            return []
        flat, offs = self._pack(pats)
        ctx = self._ctx
        hoffs = np.empty(len(pats) + 1, np.uint64)
        cap = 1 << 16
        while True:
            hits = np.empty(cap, np.uint64)
            nh = C.c_uint64(cap)
            rc = ctx.lib.tc_fm_locate(ctx.handle, self._h, _ptr(flat), _ptr(offs), len(pats),
                                      _ptr(hoffs), _ptr(hits), C.byref(nh))
            if rc == _lib.TC_ERR_CAPACITY:
                cap = int(nh.value)
                continue
            ctx._check(rc)
            break
        return [hits[int(hoffs[i]):int(hoffs[i + 1])].copy() for i in range(len(pats))]

    def info(self):
        ctx = self._ctx
        N, sig, prim = C.c_uint64(), C.c_uint32(), C.c_uint64()
        cs = np.empty(_lib.TC_MAX_SIGMA, np.int16)
        cv = np.empty(_lib.TC_MAX_SIGMA, np.uint64)
        rc = ctx.lib.tc_fm_info(self._h, C.byref(N), C.byref(sig), _ptr(cs), _ptr(cv), C.byref(prim))
        if rc != 0:
            raise TcError(rc, "tc_fm_info")
        return dict(N=int(N.value), sigma=int(sig.value), c_sym=cs[:sig.value].copy(),
                    c_val=cv[:sig.value].copy(), primary=int(prim.value))


_DEFAULT = None


def default_context():
    global _DEFAULT
    if _DEFAULT is None:
        _DEFAULT = Context(0)
    return _DEFAULT


from . import bwt, fmindex, mtf, rle  # noqa: E402,F401
