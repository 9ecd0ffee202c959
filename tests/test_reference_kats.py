"""Known-answer checks against constant data that lives in the reference tree itself.

The reference has no tests, but two of its headers ARE data: sBrotligCmdLut (705 rows,
inc/common/BrotligCommandLut.h:41-747) and sBrotligReverseBits15/9
(inc/common/BrotligReverseBits.h).  The oracle and the kernels regenerate these values
arithmetically; here they are compared row by row with the reference's literals, read from
/root/reference at test time (skipped where the reference tree is absent, e.g. on the GPU box).
Nothing from those headers is stored in this repo."""
import ctypes
import os
import re

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def test_cmd_lut_all_rows(oracle):
    text = open(os.path.join(REF, "inc/common/BrotligCommandLut.h")).read()
    body = text[text.index("sBrotligCmdLut"):]
    rows = re.findall(r"\{\s*(-?\d+),\s*(-?\d+),\s*(-?\d+),\s*(-?\d+),\s*(-?\d+),\s*(-?\d+)\s*\}", body)
    assert len(rows) == 705
    for sym, row in enumerate(rows):
        ins_extra, copy_extra, dist_code, _ctx, ins_base, copy_base = (int(x) for x in row)
        ie, ce, ib, cb = (ctypes.c_uint32() for _ in range(4))
        imp = ctypes.c_int()
        oracle.brotlig_oracle_cmd_lut(sym, ctypes.byref(ie), ctypes.byref(ce), ctypes.byref(ib), ctypes.byref(cb), ctypes.byref(imp))
        assert (ie.value, ce.value, ib.value, cb.value) == (ins_extra, copy_extra, ins_base, copy_base), sym
        if sym < 704:
            assert imp.value == (1 if dist_code == 0 else 0), sym      # distance_code 0 <=> implicit last distance


def test_kernel_length_tables_match_reference_lut():
    """The 48-entry base|extra table compiled into the kernels reproduces the LUT's columns."""
    src = open(os.path.join(os.path.dirname(__file__), "..", "brotli_g_sdk_amd", "csrc", "brotlig_kernels.h")).read()
    tab = src[src.index("kLenCodeTab[48]"):]
    tab = tab[:tab.index("};")]
    ents = [(int(b), int(e)) for b, e in re.findall(r"(\d+)u \| (\d+)u << 16", tab)]
    assert len(ents) == 48
    text = open(os.path.join(REF, "inc/common/BrotligCommandLut.h")).read()
    body = text[text.index("sBrotligCmdLut"):]
    rows = re.findall(r"\{\s*(-?\d+),\s*(-?\d+),\s*(-?\d+),\s*(-?\d+),\s*(-?\d+),\s*(-?\d+)\s*\}", body)
    ins_hi = [0, 0, 0, 0, 1, 1, 0, 2, 1, 2, 2]
    cp_hi = [0, 1, 0, 1, 0, 1, 2, 0, 2, 1, 2]
    for sym in range(704):
        ie, ce, _, _, ib, cb = (int(x) for x in rows[sym])
        cell = sym >> 6
        # the kernels' packed constants for the cell -> high code bits mapping
        assert (0x298500 >> (2 * cell)) & 3 == ins_hi[cell] and (0x262444 >> (2 * cell)) & 3 == cp_hi[cell]
        ic = ins_hi[cell] * 8 + ((sym >> 3) & 7)
        cc = cp_hi[cell] * 8 + (sym & 7)
        assert ents[ic] == (ib, ie) and ents[24 + cc] == (cb, ce), sym


def test_reverse_bits_tables():
    text = open(os.path.join(REF, "inc/common/BrotligReverseBits.h")).read()
    for name, bits in (("sBrotligReverseBits9", 9), ("sBrotligReverseBits15", 15)):
        body = text[text.index(name):]
        body = body[body.index("{") + 1:body.index("};")]
        vals = [int(v, 0) for v in re.findall(r"0x[0-9a-fA-F]+|\d+", body)]
        assert len(vals) == 1 << bits
        for i in (list(range(0, 1 << bits, 37)) + [(1 << bits) - 1]):
            assert vals[i] == int(format(i, f"0{bits}b")[::-1], 2), (name, i)
