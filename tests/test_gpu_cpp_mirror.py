"""Builds and runs the C++ host mirror's HUnit program (tests/cpp/hunit_mirror.cpp) against
libtextcomp.so on the GPU."""
import os
import subprocess
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dump(golden, path):
    def el(e):
        return "N" if e is None else "J " + e
    with open(path, "w") as f:
        for v in golden["rle"]:
            f.write("rle\n%s\n%d\n" % (v["text"], len(v["rle"])))
            f.writelines(el(e) + "\n" for e in v["rle"])
        for v in golden["mtf"]:
            f.write("mtf\n%s\n%d\n" % (v["text"], len(v["indices"])))
            f.writelines("%d\n" % i for i in v["indices"])
            f.write("%d\n" % len(v["final_list"]))
            f.writelines(el(e) + "\n" for e in v["final_list"])
        pats = [("abra", 2), ("a", 5), ("abracadabra", 1), ("x", 0), ("xra", 2), ("rab", 0)]
        f.write("count\n%s\n%d\n" % (golden["fmindex_doc"]["text"], len(pats)))
        f.writelines("%s\n%d\n" % p for p in pats)


def test_cpp_mirror_hunit(golden):
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "hunit_mirror")
        lib = os.path.join(ROOT, "text-compression_amd")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(lib, "host"), os.path.join(ROOT, "tests", "cpp", "hunit_mirror.cpp"),
                               "-o", exe, "-L", lib, "-ltextcomp", "-Wl,-rpath," + lib,
                               "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
        vec = os.path.join(d, "vectors.txt")
        _dump(golden, vec)
        out = subprocess.run([exe, vec], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "0 failures" in out.stdout
