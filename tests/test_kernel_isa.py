"""The shipped gfx950 code of the decode kernel must not touch scratch memory inside a round.

The kernel lives at the edge of its register budget (128 VGPRs for four wavefronts per SIMD); a small change in the source can move the
register allocation so that a value is spilled and reloaded once per ROUND -- 4.7 % of the kernel when it happened in round 4, and
invisible in the `-g` build that the profiling tools read their source lines from.  This test compiles the kernel the way the product does
(hipcc cross-compiles for gfx950 without a GPU) and looks at the round loops of the result."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc (cross-compiles gfx950 without a GPU)")
def test_no_scratch_access_inside_a_round():
    import isa_budget
    co = isa_budget.build([], False)                        # -O3, no -g: the code that ships
    loops = isa_budget.round_loop_scratch(co)
    outer = {lo for lo, hi, n, sc in loops if n >= 1500}
    assert len(outer) >= 2, loops                           # the two-page and the one-page instantiation of the page loop, at least
    spilled = [(lo, hi, n, sc) for lo, hi, n, sc in loops if sc]
    assert not spilled, "scratch access inside a round: %r" % spilled


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc (cross-compiles gfx950 without a GPU)")
def test_small_batch_and_deconditioning_kernels_do_not_spill():
    """The two-wavefront kernel (256 registers to spend) and the de-conditioning gather keep everything in registers."""
    import isa_budget
    co = isa_budget.build([], False)
    for name in ("brotlig_decode_duo_kernel", "brotlig_decondition_kernel"):
        sym, start, size = isa_budget.kernel_symbol(co, name)
        scratch = [i["op"] for i in isa_budget.disassemble(co, sym) if i["op"].startswith("scratch_")]
        assert not scratch, (name, scratch[:8])
