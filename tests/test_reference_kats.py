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
    src = open(os.path.join(os.path.dirname(__file__), "..", "brotli_g_sdk_amd", "csrc", "brotlig_kernel_common.h")).read()
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


# ---- block-layout constants and header bit widths ---------------------------------------------------------------
# inc/common/BrotligConstants.h holds the BC1-BC5 sub-block layout (:180-243) and the widths of the two header
# bit-fields (:47-68, :133-139); inc/common/BrotligDataConditioner.h:96-183 says which macro fills which
# sub-block slot and which slots are colour endpoints; inc/DataStream.h:28-37,:89-98 gives the field order.
# All are read from the reference tree at test time and compared with what the oracle and the kernels derive.

def _macros():
    text = open(os.path.join(REF, "inc/common/BrotligConstants.h")).read()
    vals = {}
    for name, body in re.findall(r"^#define\s+(BROTLIG_\w+)\s+(\(?-?\d+\)?)\s*$", text, re.M):
        vals[name] = int(body.strip("()"))
    return vals


def _initialize_cases():
    """format number -> (block bytes macro, block pixels macro, [sub-block size macros], [colour sub-blocks])
    parsed from the switch in BrotligDataconditionParams::Initialize."""
    text = open(os.path.join(REF, "inc/common/BrotligDataConditioner.h")).read()
    body = text[text.index("bool Initialize("):]
    body = body[:body.index("default:")]
    fmt_enum = open(os.path.join(REF, "inc/common/BrotligCommon.h")).read()
    numbers = {name: int(v, 0) for name, v in re.findall(r"(BROTLIG_DATA_FORMAT_\w+)\s*=\s*(\d+)", fmt_enum)}
    out = {}
    for m in re.finditer(r"case\s+(BROTLIG_DATA_FORMAT_BC\d):(.*?)break;", body, re.S):
        blk = m.group(2)
        sizes = [mm for _, mm in sorted((int(i), n) for i, n in re.findall(r"subBlockSizes\[(\d)\]\s*=\s*(\w+);", blk))]
        colors = [int(c) for c in re.findall(r"colorSubBlocks\[numColorSubBlocks\+\+\]\s*=\s*(\d+);", blk)]
        out[numbers[m.group(1)]] = (re.search(r"blockSizeBytes\s*=\s*(\w+);", blk).group(1),
                                    re.search(r"blockSizePixels\s*=\s*(\w+);", blk).group(1),
                                    re.search(r"numSubBlocks\s*=\s*(\w+);", blk).group(1), sizes, colors)
    return out


def _precon_words(fmt, w, h, mips=1, swizzle=0, aligned=0, pitch=None, widths=None):
    """PreconditionHeader words from the field order of inc/DataStream.h:89-98 and the macro widths."""
    order = ["SWIZZLING_BITS", "PITCH_D3D12_ALIGNED_FLAG_BITS", "TEX_WIDTH_BLOCK_BITS", "TEX_HEIGHT_BLOCK_BITS",
             "DATA_FORMAT", "TEX_NUMMIPLEVELS_BITS", "TEX_PITCH_BYTES_BITS"]
    vals = [swizzle, aligned, w - 1, h - 1, fmt, mips - 1, pitch - 1]
    word, pos = 0, 0
    for name, v in zip(order, vals):
        nb = widths["BROTLIG_PRECON_" + name]
        assert 0 <= v < (1 << nb)
        word |= v << pos
        pos += nb
    assert pos == 64
    return word & 0xFFFFFFFF, word >> 32


def _sim():
    # one builder for the simulator library (flags, dependency list, atomic replace under pytest-xdist): tests/test_sim_decode.py
    from test_sim_decode import build_sim
    return build_sim("libbrotlig_sim.so")


def test_struct_field_order_matches_datastream_h():
    """The field orders assumed above are the ones inc/DataStream.h declares."""
    text = open(os.path.join(REF, "inc/DataStream.h")).read()
    pre = text[text.index("struct PreconditionHeader"):]
    fields = re.findall(r"uint32_t\s+(\w+)\s*:\s*(BROTLIG_\w+);", pre[:pre.index("inline")])
    assert [f[1] for f in fields] == ["BROTLIG_PRECON_SWIZZLING_BITS", "BROTLIG_PRECON_PITCH_D3D12_ALIGNED_FLAG_BITS",
                                      "BROTLIG_PRECON_TEX_WIDTH_BLOCK_BITS", "BROTLIG_PRECON_TEX_HEIGHT_BLOCK_BITS",
                                      "BROTLIG_PRECON_DATA_FORMAT", "BROTLIG_PRECON_TEX_NUMMIPLEVELS_BITS",
                                      "BROTLIG_PRECON_TEX_PITCH_BYTES_BITS"]
    hdr = text[text.index("struct StreamHeader"):]
    hdr = hdr[:hdr.index("inline")]
    assert re.findall(r"(uint8_t|uint16_t)\s+(\w+);", hdr) == [("uint8_t", "Id"), ("uint8_t", "Magic"), ("uint16_t", "NumPages")]
    assert [f[1] for f in re.findall(r"uint32_t\s+(\w+)\s*:\s*(BROTLIG_\w+);", hdr)] == [
        "BROTLIG_STREAM_PAGE_SIZE_IDX_BITS", "BROTLIG_STREAM_LASTPAGE_SIZE_BITS", "BROTLIG_STREAM_PRECONDITION_BITS",
        "BROTLIG_STREAM_RESERVED_BITS"]


def test_bc_block_layouts_match_reference_macros(oracle):
    """Block bytes, sub-block sizes and colour sub-blocks of BC1..BC5: the reference's macros, the oracle's
    dc_init and the kernels' dc_init (compiled for the CPU by tests/sim) must agree."""
    M = _macros()
    cases = _initialize_cases()
    assert sorted(cases) == [1, 2, 3, 4, 5]
    oracle.brotlig_oracle_dc_layout.restype = ctypes.c_int
    oracle.brotlig_oracle_dc_layout.argtypes = [ctypes.c_uint32] * 3 + [ctypes.POINTER(ctypes.c_uint32)]
    sim = _sim()
    sim.sim_dc_table.restype = ctypes.c_int
    sim.sim_dc_table.argtypes = [ctypes.c_uint32] * 3 + [ctypes.POINTER(ctypes.c_uint32)]
    for fmt, (bb_m, px_m, nsub_m, size_ms, colors) in cases.items():
        bb, px, nsub = M[bb_m], M[px_m], M[nsub_m]
        sizes = [M[m] for m in size_ms]
        assert len(sizes) == nsub and sum(sizes) == bb
        W, H = 12, 10
        w0, w1 = _precon_words(fmt, W, H, pitch=W * bb, widths=M)
        out_size = W * H * bb
        lay = (ctypes.c_uint32 * 15)()
        assert oracle.brotlig_oracle_dc_layout(w0, w1, out_size, lay) == 1
        assert (lay[0], lay[1], lay[2]) == (bb, px, nsub), fmt
        assert list(lay[3:3 + nsub]) == sizes, fmt
        assert lay[9] == len(colors) and list(lay[10:10 + len(colors)]) == colors, fmt
        assert lay[14] == W * H
        # kernels: DcTable words (struct DcTable in csrc/brotlig_kernels.h): precon, swizzle, block_bytes, num_sub,
        # num_mips, total_blocks, tex_bytes, color_mask, sub_size[6], sub_off[6], sub_stream_off[7], ...
        t = (ctypes.c_uint32 * 256)()
        assert sim.sim_dc_table(w0, w1, out_size, t) == 1
        assert (t[2], t[3], t[4], t[5], t[6]) == (bb, nsub, 1, W * H, W * H * bb), fmt
        assert t[7] == sum(1 << c for c in colors), fmt
        assert list(t[8:8 + nsub]) == sizes, fmt
        offs = [sum(sizes[:i]) for i in range(nsub)]
        assert list(t[14:14 + nsub]) == offs, fmt
        assert list(t[20:20 + nsub + 1]) == [W * H * o for o in offs + [bb]], fmt


def test_precondition_header_bit_widths(oracle):
    """Width/height/pitch/mips/format are extracted at the bit positions the reference's macros imply."""
    M = _macros()
    assert sum(M["BROTLIG_PRECON_" + n] for n in ("SWIZZLING_BITS", "PITCH_D3D12_ALIGNED_FLAG_BITS", "TEX_WIDTH_BLOCK_BITS",
               "TEX_HEIGHT_BLOCK_BITS", "DATA_FORMAT", "TEX_NUMMIPLEVELS_BITS", "TEX_PITCH_BYTES_BITS")) == 64
    sim = _sim()
    sim.sim_dc_table.restype = ctypes.c_int
    sim.sim_dc_table.argtypes = [ctypes.c_uint32] * 3 + [ctypes.POINTER(ctypes.c_uint32)]
    oracle.brotlig_oracle_decondition_addr.restype = ctypes.c_uint32
    for (fmt, W, H, swz, pitch) in [(3, 1000, 3, 1, 16000 + 48), (1, 32767, 2, 0, 8 * 32767), (5, 7, 5000, 1, 7 * 16 + 1),
                                    (4, 1, 1, 0, 8), (2, 513, 129, 1, 513 * 16)]:
        w0, w1 = _precon_words(fmt, W, H, swizzle=swz, pitch=pitch, widths=M)
        out_size = pitch * H
        t = (ctypes.c_uint32 * 256)()
        assert sim.sim_dc_table(w0, w1, out_size, t) == 1, (fmt, W, H)
        # DcTable: w[] at word 27, h[] at 27 + kMaxMips, pitch[] after that (kMaxMips = 32)
        assert t[1] == swz
        assert t[27] == W and t[27 + 32] == H and t[27 + 64] == pitch, (fmt, W, H, list(t[24:30]))
        # the oracle reads the same fields: the last conditioned byte maps inside the texture
        assert oracle.brotlig_oracle_decondition_addr(w0, w1, out_size, 0) != 0xFFFFFFFF


def test_stream_header_bit_widths():
    M = _macros()
    widths = [M["BROTLIG_STREAM_" + n] for n in ("ID_BITS", "MAGIC_BITS", "NUM_PAGES_BITS", "PAGE_SIZE_IDX_BITS",
                                                  "LASTPAGE_SIZE_BITS", "PRECONDITION_BITS", "RESERVED_BITS")]
    assert sum(widths) == 64 and M["BROTLIG_PAGE_HEADER_NPOSTFIX_BITS"] == 2 and M["BROTLIG_PAGE_HEADER_NDIST_BITS"] == 4 \
        and M["BROTLIG_PAGE_HEADER_ISDELTAENCODED_BITS"] == 1 and M["BROTLIG_PAGE_HEADER_RESERVED_BITS"] == 1
    sim = _sim()
    sim.sim_stream_header.restype = ctypes.c_int
    sim.sim_stream_header.argtypes = [ctypes.c_uint32] * 2 + [ctypes.POINTER(ctypes.c_uint32)]
    for (pages, idx, last, pre) in [(1, 0, 1, 0), (65535, 1, 0, 1), (4096, 2, 131071, 0), (3, 1, 65535, 1)]:
        vals = [5, 5 ^ 0xFF, pages, idx, last, pre, 0]
        word, pos = 0, 0
        for nb, v in zip(widths, vals):
            assert v < (1 << nb)
            word |= v << pos
            pos += nb
        out = (ctypes.c_uint32 * 6)()
        assert sim.sim_stream_header(word & 0xFFFFFFFF, word >> 32, out) == 1
        page = 32768 << idx
        assert list(out[:5]) == [pages, page, last, pre, 16 if pre else 8]
        assert out[5] == (pages * page - (page - last if last else 0)) & 0xFFFFFFFF


def test_precondition_header_rejects_wrapping_geometry():
    """A header whose 32-bit products wrap to exactly out_size (h = 32768, pitch = 393218, w = 1: pitch * h =
    2^32 + 65536) must be refused by the kernels' dc_init: accepted, the de-conditioning kernel would walk
    32768 real rows past the output."""
    M = _macros()
    sim = _sim()
    sim.sim_dc_table.restype = ctypes.c_int
    sim.sim_dc_table.argtypes = [ctypes.c_uint32] * 3 + [ctypes.POINTER(ctypes.c_uint32)]
    w0, w1 = _precon_words(3, 1, 32768, pitch=393218, widths=M)
    t = (ctypes.c_uint32 * 256)()
    assert (393218 * 32768) & 0xFFFFFFFF == 65536
    assert sim.sim_dc_table(w0, w1, 65536, t) == 0
