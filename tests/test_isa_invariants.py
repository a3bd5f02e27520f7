"""The hand-placed counted LDS waits of the 16-bit network kernels are only valid under invariants of the generated
code (tools/check_isa.py).  Checked on the objects `__graft_entry__.build()` leaves under csrc/build; skipped when the
library was not built in this checkout (e.g. the GPU box, which only receives the .so)."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(REPO, "nonrigid_nerf_amd", "csrc", "build")


@pytest.mark.skipif(not os.path.isdir(BUILD) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"),
                    reason="no build directory / no llvm-objdump")
def test_network_kernels_keep_the_wait_invariants():
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import check_isa
    errors = check_isa.check(BUILD)
    assert not errors, "\n".join(errors)
