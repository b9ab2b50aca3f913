"""The workspace of a context under repeated growth and release (round 4; VERDICT round 3, weak 8 / ADVICE medium 1).

Round 3 made the workspace of a long record a range of separately created physical chunks (HIP virtual memory
management) and saw ONE GPU memory fault after several such workspaces had been created and released in a row in
one process.  Since round 4 a chunked workspace GROWS in place (more chunks mapped into its reserved address range;
nothing is unmapped while the context lives) and is released only with its context, after a device-wide
synchronisation.  These tests walk exactly the paths that used to release: a context that meets longer and longer
records, many contexts created and destroyed, and chunked workspaces of two sizes created and released 30 times --
each in a child process, so that a fault is a failed test and not a dead session.  Every encode is checked by its
round trip on the device."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(%(root)r, "text-compression_amd"))
import torch, textcomp
from textcomp import Block
mode = sys.argv[1]

def roundtrip(ctx, n, seed):
    lib = ctx.lib
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert lib.tc_generate_dev(ctx.handle, 0, seed, n, C.c_void_p(t.data_ptr())) == 0
    cnt = torch.empty(n + 2, dtype=torch.int32, device="cuda"); val = torch.empty(n + 2, dtype=torch.int16, device="cuda")
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    blk = Block(); blk.nruns = n + 2; blk.run_count = cnt.data_ptr(); blk.run_value = val.data_ptr()
    rc = lib.tc_encode_dev(ctx.handle, C.c_void_p(t.data_ptr()), n, C.byref(blk))
    assert rc == 0, lib.tc_last_error(ctx.handle)
    rc = lib.tc_decode_dev(ctx.handle, C.byref(blk), C.c_void_p(out.data_ptr()))
    assert rc == 0, lib.tc_last_error(ctx.handle)
    torch.cuda.synchronize()
    assert torch.equal(out, t), "round trip of %%d bytes" %% n
    return ctx.stats()

if mode == "grow":
    # ONE context, records of 2^29 and 2^30 bytes interleaved: the second forces the workspace to grow
    with textcomp.Context(0) as ctx:
        st = roundtrip(ctx, 1 << 29, 1)
        c0 = st.ws_chunks
        assert c0 > 0, "a record of 2^29 bytes should get a chunked workspace"
        st = roundtrip(ctx, 1 << 30, 2)
        assert st.ws_chunks > c0 and st.ws_grown == 1, (c0, st.ws_chunks, st.ws_grown)
        for i, lg in enumerate((29, 30, 29, 30)):
            st = roundtrip(ctx, 1 << lg, 3 + i)
        assert st.ws_grown == 1          # (nothing shrinks, nothing is placed again)
elif mode == "contexts":
    # ten contexts created and destroyed, each with a chunked workspace (release path: context end)
    for i in range(10):
        with textcomp.Context(0) as ctx:
            st = roundtrip(ctx, 1 << 29, 10 + i)
            assert st.ws_chunks > 0
elif mode == "cycle":
    # chunked workspaces of two sizes created and released 30 times (small records: TC_WS_VMM_MIN_LOG2 lowered),
    # a second context alive all the while with work of its own in flight on its stream
    with textcomp.Context(0) as other:
        for i in range(30):
            with textcomp.Context(0) as ctx:
                st = roundtrip(ctx, (1 << 24) if i %% 2 else (1 << 25), 40 + i)
                assert st.ws_chunks > 0
                st = roundtrip(ctx, 1 << 26, 80 + i)          # grows in place
                assert st.ws_grown == 1
            roundtrip(other, 1 << 22, 120 + i)
print("ok", mode)
'''


@pytest.mark.parametrize("mode,env", [("grow", {}), ("contexts", {}),
                                      ("cycle", {"TC_WS_VMM_MIN_LOG2": "20", "TC_WS_VMM": "26"})])
def test_workspace_stress(mode, env):
    e = dict(os.environ)
    e.update(env)
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, mode], env=e, cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = (p.stdout or "")[-1500:] + (p.stderr or "")[-2500:]
    assert p.returncode == 0 and ("ok %s" % mode) in p.stdout, tail
