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
    """The de-conditioning kernel keeps everything in registers (59 of them: eight wavefronts per SIMD).  The two-wavefront kernel is held to 128
    registers (round 5: `__launch_bounds__(128, 4)` -- its LDS allows eight workgroups per compute unit, and the new table builder had taken
    145 registers, i.e. six: 2 048 pages 1.10 against 0.83 ms); at 128 it keeps a handful of page-scope values in scratch -- three stores at page
    start, three reloads -- and no more than that."""
    import isa_budget
    co = isa_budget.build([], False)
    sym, start, size = isa_budget.kernel_symbol(co, "brotlig_decondition_kernel")
    scratch = [i["op"] for i in isa_budget.disassemble(co, sym) if i["op"].startswith("scratch_")]
    assert not scratch, ("brotlig_decondition_kernel", scratch[:8])
    sym, start, size = isa_budget.kernel_symbol(co, "brotlig_decode_duo_kernel")
    scratch = [i["op"] for i in isa_budget.disassemble(co, sym) if i["op"].startswith("scratch_")]
    assert len(scratch) <= 8, ("brotlig_decode_duo_kernel", scratch)
    # ... and WHERE they sit: none inside a round of either wavefront -- the loops of a few hundred instructions and more that take no
    # page from the counter are the producer's and the consumer's round and group loops
    loops = isa_budget.loops_scratch(co, "brotlig_decode_duo_kernel")
    assert len(loops) >= 2, loops
    assert not [l for l in loops if l[3]], [l for l in loops if l[3]]


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc (cross-compiles gfx950 without a GPU)")
def test_no_kernel_divides_through_one_float_reciprocal():
    """Round 5: `(hash >> 8) % supers` in the de-conditioning kernel had two operands below 2^24; this toolchain lowers such a division to a
    single float reciprocal (v_rcp_iflag_f32, v_mul_f32, v_trunc_f32) with a correction for a quotient that is too SMALL only.  13258079 / 11
    comes out one too large, the remainder was -1 & 0xFFFFFF, a table walk ran off its table: a GPU memory fault for one texture in ten
    thousand (the simulator divides exactly).  The expression is a multiply-high now; this test keeps the lowering out of every kernel:
    its signature is v_trunc_f32 (the exact 32-bit lowering and the kernels' own div_small / mod_u16 do not use it)."""
    import isa_budget
    co = isa_budget.build([], False)
    text = isa_budget.sh([f"{isa_budget.LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co])
    assert "v_rcp_iflag_f32" in text            # (the listing is what it should be: the exact lowerings are there)
    assert "v_trunc_f32" not in text, "a kernel divides through one float reciprocal: check the operands' ranges and avoid the division"


def test_the_float_reciprocal_division_is_inexact_for_24_bit_operands():
    """The arithmetic of the lowering above, restated: trunc(float(a) * float(1 / b)) for a = 13258079, b = 11 is one too large."""
    import numpy as np
    a, b = np.float32(13258079), np.float32(11)
    q = np.trunc(a * (np.float32(1) / b))
    assert int(q) == 13258079 // 11 + 1


def test_stage_budget_tool_finds_its_markers_in_the_split_headers():
    """profiles/tools/isa_budget.py books every instruction of the page loop to a stage of a round by marker comments in the source of
    decode_pages<> -- which moved from brotlig_kernels.h to brotlig_round.h when the header was split per stage (round 6), and the tool stopped
    working unnoticed.  No compiler needed: the markers must all be there, in order."""
    import isa_budget
    src = open(os.path.join(isa_budget.CSRC, "brotlig_round.h")).read().split("\n")
    fn_line, marks = isa_budget.stage_table(src)
    lines = [ln for _, ln in marks]
    assert fn_line < lines[0] and lines == sorted(lines) and len(set(lines)) == len(lines), marks
