"""Checks on the built device code that need no GPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_compiler_use_of_m0_in_the_dma_kernels():
    """The LDS-DMA code changes M0 behind the compiler's back (wn_dma.h); the built detection and
    sync kernels must contain no other instruction that touches it."""
    import check_m0
    objs = [os.path.join(ROOT, "wavenet_amd", "csrc", f) for f in ("wn_detect.o", "wn_sync.o")]
    if not all(os.path.exists(o) for o in objs) or not os.path.exists(os.path.join(check_m0.LLVM, "llvm-objdump")):
        pytest.skip("object files or llvm-objdump not present")
    for obj in objs:
        ours, bad = check_m0.check(obj)
        assert ours > 0 and not bad, bad[:5]
