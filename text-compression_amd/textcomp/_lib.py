"""ctypes binding of libtextcomp.so (include/textcomp.h).  No CPU fallback: if the
HIP library is missing or no MI355X is usable, calls raise."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TEXTCOMP_LIB") or os.path.join(os.path.dirname(_HERE), "libtextcomp.so")   # (TEXTCOMP_LIB: a variant build, for A/B runs)

TC_OK = 0
TC_ERR_ARG, TC_ERR_CAPACITY, TC_ERR_MALFORMED, TC_ERR_HIP, TC_ERR_OOM, TC_ERR_INTERNAL, TC_ERR_NCCL = (
    -1, -2, -3, -4, -5, -6, -7)
TC_COMM_ID_BYTES = 128
TC_MAX_SIGMA = 257
TC_MAX_ROUNDS = 40

ERR_NAMES = {-1: "TC_ERR_ARG", -2: "TC_ERR_CAPACITY", -3: "TC_ERR_MALFORMED", -4: "TC_ERR_HIP",
             -5: "TC_ERR_OOM", -6: "TC_ERR_INTERNAL", -7: "TC_ERR_NCCL"}


class TcError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "?"), code, msg))
        self.code = code


class TcMalformed(TcError):
    """Input on which the reference itself throws (fromJust / DS.index / read)."""


class Stats(C.Structure):
    _fields_ = [("n", C.c_uint64), ("N", C.c_uint64), ("sigma", C.c_uint32), ("rounds", C.c_uint32),
                ("m", C.c_uint64 * TC_MAX_ROUNDS), ("key_bytes", C.c_uint32 * TC_MAX_ROUNDS),
                ("passes", C.c_uint32 * TC_MAX_ROUNDS), ("h", C.c_uint32 * TC_MAX_ROUNDS),
                ("runs", C.c_uint64), ("ms_sa", C.c_float), ("ms_bwt", C.c_float),
                ("ms_mtf", C.c_float), ("ms_rle", C.c_float), ("ms_total", C.c_float),
                ("radix_launches", C.c_uint32), ("ms_radix", C.c_float),
                ("keygen_fused", C.c_uint32), ("finish_pass", C.c_uint32),
                ("sample_dups", C.c_uint32), ("msd_path", C.c_uint32), ("msd_keyonly", C.c_uint32),
                ("ticket_fallbacks", C.c_uint32), ("ws_chunks", C.c_uint32), ("ws_grown", C.c_uint32),
                ("seg_rounds", C.c_uint32), ("chain_rounds", C.c_uint32)]


class Block(C.Structure):
    _fields_ = [("n", C.c_uint64), ("primary", C.c_uint64), ("sigma", C.c_uint32),
                ("final_list", C.c_int16 * TC_MAX_SIGMA), ("nruns", C.c_uint64),
                ("run_count", C.c_void_p), ("run_value", C.c_void_p)]


# every symbol include/textcomp.h declares: (name, restype, argtypes)
_P, _U64, _I64, _U32, _INT = C.c_void_p, C.c_uint64, C.c_int64, C.c_uint32, C.c_int
_PU64 = C.POINTER(C.c_uint64)
_PU32 = C.POINTER(C.c_uint32)
SYMBOLS = [
    ("tc_ctx_create", _INT, [_INT, C.POINTER(_P)]),
    ("tc_ctx_destroy", None, [_P]),
    ("tc_last_error", C.c_char_p, [_P]),
    ("tc_version", C.c_char_p, []),
    ("tc_get_stats", _INT, [_P, C.POINTER(Stats)]),
    ("tc_ctx_stream", _P, [_P]),
    ("tc_ctx_set_profile", _INT, [_P, _INT]),
    ("tc_ctx_place_workspace", _INT, [_P, _P, _U64, _P, _INT, _P, _P]),
    ("tc_bwt_encode", _INT, [_P, _P, _U64, _P, _PU64]),
    ("tc_bwt_encode_dev", _INT, [_P, _P, _U64, _P, _PU64]),
    ("tc_suffix_array", _INT, [_P, _P, _U64, _P]),
    ("tc_bwt_decode", _INT, [_P, _P, _U64, _U64, _P]),
    ("tc_bwt_decode_sym", _INT, [_P, _P, _U64, _P, _PU64]),
    ("tc_mtf_encode", _INT, [_P, _P, _U64, _I64, _P, _P, _PU32]),
    ("tc_mtf_encode_sym", _INT, [_P, _P, _U64, _P, _P, _PU32]),
    ("tc_mtf_decode", _INT, [_P, _P, _U64, _P, _U32, _P]),
    ("tc_rle_encode", _INT, [_P, _P, _U64, _I64, _P, _P, _PU64]),
    ("tc_rle_encode_sym", _INT, [_P, _P, _U64, _P, _P, _PU64]),
    ("tc_rle_encode_u16", _INT, [_P, _P, _U64, _P, _P, _PU64]),
    ("tc_rle_decode", _INT, [_P, _P, _P, _U64, _P, _PU64]),
    ("tc_rle_decode_u16", _INT, [_P, _P, _P, _U64, _P, _PU64]),
    ("tc_encode", _INT, [_P, _P, _U64, C.POINTER(Block)]),
    ("tc_encode_dev", _INT, [_P, _P, _U64, C.POINTER(Block)]),
    ("tc_decode", _INT, [_P, C.POINTER(Block), _P]),
    ("tc_decode_dev", _INT, [_P, C.POINTER(Block), _P]),
    ("tc_block_packed_bound", _U64, [_U64, _U32]),
    ("tc_block_pack_dev", _INT, [_P, C.POINTER(Block), _P, _PU64, _PU64]),
    ("tc_block_unpack_dev", _INT, [_P, _P, _U64, _U64, _U32, _U64, C.POINTER(Block)]),
    ("tc_container_bound", _U64, [_U64, _U32]),
    ("tc_block_to_container_dev", _INT, [_P, C.POINTER(Block), _P, _PU64]),
    ("tc_encode_container_dev", _INT, [_P, _P, _U64, _P, _PU64]),
    ("tc_container_to_block_dev", _INT, [_P, _P, _U64, C.POINTER(Block)]),
    ("tc_encode_container", _INT, [_P, _P, _U64, _P, _PU64]),
    ("tc_container_info", _INT, [_P, _P, _U64, _PU64, _PU64]),
    ("tc_decode_container", _INT, [_P, _P, _U64, _P, _PU64]),
    ("tc_stream_bound", _U64, [_U64, _U64]),
    ("tc_encode_stream", _INT, [_P, _P, _U64, _U64, _P, _PU64]),
    ("tc_stream_info", _INT, [_P, _P, _U64, _PU64, _PU64]),
    ("tc_decode_stream", _INT, [_P, _P, _U64, _P, _PU64]),
    ("tc_fm_build", _INT, [_P, _P, _U64, C.POINTER(_P)]),
    ("tc_fm_build_dev", _INT, [_P, _P, _U64, C.POINTER(_P)]),
    ("tc_fm_free", None, [_P]),
    ("tc_fm_count", _INT, [_P, _P, _P, _P, _U64, _P]),
    ("tc_fm_count_dev", _INT, [_P, _P, _P, _P, _U64, _P]),
    ("tc_fm_locate", _INT, [_P, _P, _P, _P, _U64, _P, _P, _PU64]),
    ("tc_fm_info", _INT, [_P, _PU64, _PU32, _P, _P, _PU64]),
    ("tc_comm_unique_id", _INT, [_P, _P]),
    ("tc_comm_create", _INT, [_P, _P, _INT, _INT, C.POINTER(_P)]),
    ("tc_comm_destroy", None, [_P]),
    ("tc_comm_gather", _INT, [_P, _INT, _P, _U64, _P, _U64, _PU64]),
    ("tc_comm_wait", _INT, [_P]),
    ("tc_comm_reserved_cus", _INT, [_P]),
    ("tc_comm_broadcast", _INT, [_P, _INT, _P, _U64]),
    ("tc_fm_export_bound", _U64, [_P, _INT]),
    ("tc_fm_export_dev", _INT, [_P, _P, _INT, _P, _PU64]),
    ("tc_fm_import_dev", _INT, [_P, _P, _U64, C.POINTER(_P)]),
    ("tc_generate_dev", _INT, [_P, _INT, _U64, _U64, _P]),
]

_LIB = None


def _prefer_process_hip_runtime():
    """One process, one HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7;
    libtextcomp.so names the same SONAME.  Whichever copy is loaded first serves both, and
    torch does not come up on the system copy ("No HIP GPUs are available").  So, when a
    torch installation is present and its runtime is not in the process yet, load torch's
    copy first (without importing torch); TEXTCOMP_SYSTEM_HIP=1 keeps /opt/rocm's."""
    if os.environ.get("TEXTCOMP_SYSTEM_HIP") == "1":
        return
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # torch already brought its runtime in
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen libtextcomp.so and type every entry point; raises if it is missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libtextcomp.so not built (%s): run __graft_entry__.build() or `make -C "
                "text-compression_amd`; there is no CPU fallback" % LIB_PATH)
        _prefer_process_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB
