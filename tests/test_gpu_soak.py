"""Bounded, seeded slices of the long randomized runs (tests/long/fuzz_*.py: differential against the CPU oracle)
under `-m gpu`, so that the driver observes them: default paths and the path selectors that force the MSD round 0,
the timestamp MTF and the dense ranks by regions at small sizes.  Each slice is a child
process (its own context; the selectors are read from the environment at run time) and a few seconds."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LONG = os.path.join(ROOT, "tests", "long")

SLICES = [
    ("fuzz_long.py", ["150", "101", "60000"], {}),
    ("fuzz_long.py", ["80", "102", "200000"], {"TC_SA_MSD": "2", "TC_SA_MSD_MIN_LOG2": "10"}),
    ("fuzz_long.py", ["60", "103", "200000"], {"TC_SA_MSD": "2", "TC_SA_MSD_MIN_LOG2": "10", "TC_SA_MSD_BIG": "1"}),
    ("fuzz_long.py", ["80", "104", "120000"], {"TC_MTF_TS": "2", "TC_SA_BIN_MIN_LOG2": "0"}),
    ("fuzz_long.py", ["60", "105", "120000"], {"TC_SA_BIN_MIN_LOG2": "0", "TC_SA_DENSE": "1"}),
    # round 4: the segmented sort of the doubling rounds at every size (sparse and dense ranks), the streamed key
    # directory, the key round of the MSD way's whole buckets
    ("fuzz_long.py", ["80", "109", "120000"], {"TC_SA_SEG_MIN": "1", "TC_SA_ACCEL_MIN": "1"}),
    ("fuzz_long.py", ["60", "110", "200000"], {"TC_SA_SEG_MIN": "1", "TC_SA_DENSE": "1", "TC_SA_BIN_MIN_LOG2": "0"}),
    ("fuzz_long.py", ["60", "111", "300000"], {"TC_SA_MSD": "2", "TC_SA_MSD_MIN_LOG2": "10", "TC_SA_MSD_BIG": "1", "TC_SA_SEG_MIN": "1", "TC_SA_ACCEL_MIN": "1"}),
    # round 4, second half: a chain round (tc_chain.hpp) in every dense doubling round with h >= 4 -- texts made of periods
    # (fuzz_chain.py) and the general mix
    ("fuzz_chain.py", ["150", "112", "30000"], {"TC_SA_CHAIN": "2", "TC_SA_DENSE": "1", "TC_SA_SEG_MIN": "1"}),
    ("fuzz_chain.py", ["80", "113", "120000"], {"TC_SA_CHAIN": "2", "TC_SA_DENSE": "1", "TC_SA_SEG_MIN": "1", "TC_SA_BIN_MIN_LOG2": "0"}),
    ("fuzz_long.py", ["60", "114", "60000"], {"TC_SA_CHAIN": "2", "TC_SA_DENSE": "1", "TC_SA_SEG_MIN": "1"}),
    # ... and with SPARSE ranks (flags from the members through the rank table / the sorted keys; row blocks without a
    # position on path skipped), also behind the MSD round 0
    ("fuzz_chain.py", ["120", "115", "60000"], {"TC_SA_CHAIN": "2", "TC_SA_SEG_MIN": "1", "TC_SA_ACCEL_MIN": "1"}),
    ("fuzz_long.py", ["60", "116", "200000"], {"TC_SA_CHAIN": "2", "TC_SA_SEG_MIN": "1", "TC_SA_MSD": "2", "TC_SA_MSD_MIN_LOG2": "10"}),
    ("fuzz_raw.py", ["150", "106"], {}),
    ("fuzz_raw.py", ["100", "107"], {"TC_MTF_TS": "2"}),
    ("fuzz_fm.py", ["60", "108"], {}),
]


@pytest.mark.parametrize("script,args,env", SLICES, ids=["%s-%s-%s" % (s[0][:-3], s[1][1], "+".join(sorted(s[2])) or "default") for s in SLICES])
def test_soak_slice(script, args, env):
    e = dict(os.environ)
    e.update(env)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    p = subprocess.run([sys.executable, os.path.join(LONG, script)] + args, env=e, cwd=ROOT, capture_output=True, text=True, timeout=300)
    tail = (p.stdout or "")[-1500:] + (p.stderr or "")[-1500:]
    assert p.returncode == 0, tail
    assert "0 failures" in p.stdout, tail
