"""CPU-only: the C-ABI library loads and exports every symbol include/textcomp.h
declares; the Python binding types each of them; no compute without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(names=("textcomp.h",)):
    out = set()
    for nm in names:
        src = open(os.path.join(ROOT, "include", nm)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        out |= set(re.findall(r"\b(tc_[a-z0-9_]+)\s*\(", src))
    return sorted(out)


def test_header_symbols_exported():
    import __graft_entry__
    __graft_entry__.build()
    so = os.path.join(ROOT, "text-compression_amd", "libtextcomp.so")
    lib = ctypes.CDLL(so)
    names = _declared(sorted(os.listdir(os.path.join(ROOT, "include"))))   # every header in include/
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "libtextcomp.so lacks %s" % n


def test_binding_covers_header():
    from textcomp import _lib
    assert sorted(n for n, _, _ in _lib.SYMBOLS) == _declared()
    _lib.load()


def test_fails_loudly_without_gpu():
    """No CPU fallback: without a usable device the product path raises."""
    import textcomp
    lib = textcomp._lib.load()
    h = ctypes.c_void_p()
    rc = lib.tc_ctx_create(0, ctypes.byref(h))
    if rc == 0:
        lib.tc_ctx_destroy(h)
        pytest.skip("a GPU is present")
    assert rc == textcomp._lib.TC_ERR_HIP
    with pytest.raises(textcomp.TcError):
        textcomp.Context(0)
    with pytest.raises(textcomp.TcError):
        textcomp.bwt.bytestringToBWT(b"abc")


def test_product_never_imports_oracle():
    """The product path must not route through the CPU oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "text-compression_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".hs")):
                txt = open(os.path.join(d, f)).read().lower()
                assert "oracle" not in txt, (d, f)


def _c_prototypes():
    """name -> number of parameters, from include/textcomp.h"""
    src = open(os.path.join(ROOT, "include", "textcomp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(tc_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        params = m.group(2).strip()
        out[m.group(1)] = 0 if params in ("", "void") else params.count(",") + 1
    return out


def test_haskell_ffi_imports_match_header():
    """hs/.../FFI.hs cannot be compiled here (no GHC): at least every `foreign import` names a function
    the header declares, with as many arguments, and GPU.hs calls only imported names."""
    hs = os.path.join(ROOT, "text-compression_amd", "hs", "Data", "TextCompression")
    ffi = open(os.path.join(hs, "FFI.hs")).read()
    protos = _c_prototypes()
    imports = re.findall(r'foreign import ccall (?:safe|unsafe) "(tc_[a-z0-9_]+)"\s*\n?\s*(c_tc_[a-z0-9_]+)\s*::\s*([^\n]+(?:\n\s{4,}[^\n]+)*)', ffi)
    assert len(imports) >= 20
    assert len(imports) == ffi.count("foreign import")
    for cname, hname, sig in imports:
        assert hname == "c_" + cname
        assert cname in protos, "FFI.hs imports %s, which include/textcomp.h does not declare" % cname
        assert sig.count("->") == protos[cname], (cname, sig, protos[cname])
    gpu = open(os.path.join(hs, "GPU.hs")).read()
    used = set(re.findall(r"\bc_tc_[a-z0-9_]+", gpu))
    assert used and used <= {h for _, h, _ in imports}, used - {h for _, h, _ in imports}
    # the ByteString surface is complete: these too
    for fn in ("bytestringFMIndexLocateS", "bytestringFMIndexLocateP", "bytestringFromByteStringBWT",
               "bytestringToBWTToFMIndexB"):
        assert re.search(r"^%s ::" % fn, gpu, flags=re.M), fn
    assert "NOINLINE theCtx" in gpu      # one process-global context


def test_integration_md_excerpt_is_ffi_hs():
    """INTEGRATION.md shows the Haskell bindings a maintainer would add: the excerpt is FFI.hs itself, and every
    other Haskell snippet of the document uses only names FFI.hs imports."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    ffi = open(os.path.join(ROOT, "text-compression_amd", "hs", "Data", "TextCompression", "FFI.hs")).read()
    m = re.search(r"<!-- FFI.hs begin -->\n```haskell\n(.*?)\n```\n<!-- FFI.hs end -->", md, flags=re.S)
    assert m, "INTEGRATION.md lost its FFI.hs excerpt markers"
    assert m.group(1).strip() == ffi.strip(), "INTEGRATION.md's excerpt and hs/.../FFI.hs differ: regenerate the excerpt"
    imported = set(re.findall(r"\bc_tc_[a-z0-9_]+", ffi))
    for name in set(re.findall(r"\bc_[a-z0-9_]+", md)):
        assert name in imported, "INTEGRATION.md mentions %s, which FFI.hs does not import" % name
    assert "module Data.TextCompression.FFI" in md and "Data.TextComp.FFI" not in md
