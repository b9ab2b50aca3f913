"""CPU-only: the asm-issued prefetch of the partition kernels (csrc/tc_msd.hpp) in the ISA of the default build AND of the
diagnostic variant builds scripts/README.md documents (their extra code changes the register allocation): no instruction
between the prefetch loads and the landing wait may touch a destination register.  The check itself is
scripts/check_asm_prefetch.py (also run by `make variant` for every variant .so)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("check_asm_prefetch", os.path.join(ROOT, "scripts", "check_asm_prefetch.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)


@pytest.mark.parametrize("defs", [[], ["-DMSD_PROFILE"], ["-DMSDK_PROFILE"], ["-DMTFRLE_PROFILE"]],
                         ids=["default", "MSD_PROFILE", "MSDK_PROFILE", "MTFRLE_PROFILE"])
def test_prefetch_destinations_are_untouched_until_they_land(defs):
    assert _mod.check(defs) >= 4
