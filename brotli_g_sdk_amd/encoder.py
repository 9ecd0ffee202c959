"""ctypes binding of the functional Brotli-G encoder (input generator; the role of
BrotliG::Encode, inc/BrotligEncoder.h:34-37)."""
import ctypes

import numpy as np

from . import _build

NO_CODELEN_RLE = 1 << 0
FORCE_STORED = 1 << 1
NO_RING_CODES = 1 << 2
NO_LAZY = 1 << 3
LITERALS_ONLY = 1 << 4
FORCE_COMPLEX_TABLES = 1 << 5
SEARCH_DIST_PARAMS = 1 << 6      # per page: NPOSTFIX / NDIRECT chosen by estimated distance cost
OPTIMAL_PARSE = 1 << 7           # shortest-path parse under the symbol costs of a first (lazy) parse
DECODER_CORNERS = 1 << 9         # what the reference's decoder accepts and its encoder never writes (code-length tokens, reserved bits, IS_DELTA on plain pages, simple-code order)
SMOOTH_HISTOGRAMS = 1 << 8       # smoothed symbol counts for the prefix codes where that makes the page smaller

FORMAT_BC1, FORMAT_BC2, FORMAT_BC3, FORMAT_BC4, FORMAT_BC5 = 1, 2, 3, 4, 5


class _Options(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in (
        "page_size", "npostfix", "ndirect_m", "flags", "max_chain",
        "precondition", "swizzle", "delta", "format",
        "width_blocks", "height_blocks", "num_mips", "pitch_bytes", "pitch_d3d12_aligned", "num_threads")]


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build_encoder())
        _lib.BrotligEncMaxCompressedSize.restype = ctypes.c_uint32
        _lib.BrotligEncMaxCompressedSize.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        _lib.BrotligEncode.restype = ctypes.c_int
        _lib.BrotligEncode.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32),
                                       ctypes.c_void_p, ctypes.POINTER(_Options)]
    return _lib


def encode(data, page_size=65536, npostfix=0, ndirect_m=0, flags=0, max_chain=0, precondition=None, threads=0) -> np.ndarray:
    """Encode `data` (bytes-like / uint8 array) into one .brotlig stream (uint8 array).

    precondition: None, or a dict(format=1..5, width_blocks, height_blocks, num_mips=1, swizzle=False,
    delta=False, pitch_bytes=0, pitch_d3d12_aligned=False); a `page_size` key in it overrides the argument (the test
    cases carry their page size with the texture's description).
    """
    lib = _load()
    if precondition and precondition.get("page_size"):
        page_size = precondition["page_size"]
    src = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data,
                               dtype=np.uint8)
    o = _Options(page_size=page_size, npostfix=npostfix, ndirect_m=ndirect_m, flags=flags, max_chain=max_chain,
                 num_threads=threads)                                 # 0 = one worker per hardware thread (pages are independent)
    if precondition:
        o.precondition = 1
        o.format = precondition["format"]
        o.width_blocks = precondition["width_blocks"]
        o.height_blocks = precondition["height_blocks"]
        o.num_mips = precondition.get("num_mips", 1)
        o.swizzle = int(bool(precondition.get("swizzle", False)))
        o.delta = int(bool(precondition.get("delta", False)))
        o.pitch_bytes = precondition.get("pitch_bytes", 0)
        o.pitch_d3d12_aligned = int(bool(precondition.get("pitch_d3d12_aligned", False)))
    cap = lib.BrotligEncMaxCompressedSize(len(src), page_size)
    out = np.empty(cap, dtype=np.uint8)
    osz = ctypes.c_uint32(cap)
    rc = lib.BrotligEncode(len(src), src.ctypes.data, ctypes.byref(osz), out.ctypes.data, ctypes.byref(o))
    if rc != 0:
        raise ValueError(f"BrotligEncode failed with code {rc}")
    return out[:osz.value].copy()
