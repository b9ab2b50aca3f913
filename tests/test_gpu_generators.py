"""tc_generate_dev kinds 2 .. 6 (the non-iid records of bench.py's classes leg) against their numpy restatements in
tests/classgen.py: every byte is the same integer function of (kind, seed, position) on both sides, which is what lets the
CPU oracle encode exactly what the device generates (tests/golden/classes_digest.json: dev_periodic)."""
import ctypes as C

import numpy as np
import pytest

import classgen

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(classgen.DEV_KINDS))
@pytest.mark.parametrize("n", [1, 4095, 4097, (1 << 22) + 123])
def test_device_generator_equals_numpy(name, n):
    import torch
    import textcomp
    kind, seed = classgen.DEV_KINDS[name]
    with textcomp.Context(0) as ctx:
        d = torch.empty(n, dtype=torch.uint8, device="cuda")
        assert ctx.lib.tc_generate_dev(ctx.handle, kind, seed, n, C.c_void_p(d.data_ptr())) == 0
        got = d.cpu().numpy()
    want = classgen.make(name, n)
    assert np.array_equal(got, want), "first difference at %d" % int(np.nonzero(got != want)[0][0])
