"""The drop-in boundary: libbrotlig_hip.so must load (no GPU needed to load it) and export every
function include/brotlig_amd.h declares.  No compute calls here."""
import ctypes
import os
import re

import numpy as np

from brotli_g_sdk_amd import _build
from helpers import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "brotlig_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text)))


def test_library_exports_every_declared_symbol():
    so = ctypes.CDLL(_build.build_hip())
    names = declared_functions()
    assert {"DecompressedSize", "DecodeGPU", "BrotligDecodeBatchDevice", "BrotligDecodeBatchStatus",
            "BrotligDecodeBatchTimed", "BrotligDecodeWorkspaceSize", "BrotligDeviceSelfTest"} <= set(names)
    for n in names:
        assert hasattr(so, n), n


def test_decompressed_size_needs_no_device():
    from brotli_g_sdk_amd import api
    hdr = np.frombuffer(bytes.fromhex("05fa0300a10f0000"), dtype=np.uint8)      # 3 pages, 64 KiB, last = 1000
    assert api.DecompressedSize(hdr) == 2 * 65536 + 1000


def test_workspace_size_grows_with_streams():
    so = ctypes.CDLL(_build.build_hip())
    so.BrotligDecodeWorkspaceSize.restype = ctypes.c_size_t
    so.BrotligDecodeWorkspaceSize.argtypes = [ctypes.c_uint32]
    assert so.BrotligDecodeWorkspaceSize(4096) > so.BrotligDecodeWorkspaceSize(1) >= 1024
    # with room for the page schedule: one word per 32 KiB page the output could hold, plus one per stream
    so.BrotligDecodeWorkspaceSizeFor.restype = ctypes.c_size_t
    so.BrotligDecodeWorkspaceSizeFor.argtypes = [ctypes.c_uint32, ctypes.c_uint64]
    base = so.BrotligDecodeWorkspaceSize(16)
    assert so.BrotligDecodeWorkspaceSizeFor(16, 0) >= base
    assert so.BrotligDecodeWorkspaceSizeFor(16, 4 << 30) >= base + 4 * ((4 << 30) // 32768)


def test_streamer_refuses_bad_arguments_without_a_device():
    """Argument checks of the streaming front end come before any HIP call."""
    so = ctypes.CDLL(_build.build_hip())
    so.BrotligStreamerCreate.restype = ctypes.c_int
    so.BrotligStreamerCreate.argtypes = [ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p]
    h = ctypes.c_void_p()
    assert so.BrotligStreamerCreate(0, 1 << 20, 1 << 20, 16, ctypes.byref(h)) != 0          # no slots
    assert so.BrotligStreamerCreate(3, 1 << 20, 1 << 20, 5000, ctypes.byref(h)) != 0       # > 4096 streams per batch
    assert so.BrotligStreamerCreate(3, 1 << 20, 1 << 20, 16, None) != 0
    so.BrotligStreamerWait.restype = ctypes.c_int
    so.BrotligStreamerWait.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    assert so.BrotligStreamerWait(None, 1) != 0


def test_product_does_not_reference_the_oracle():
    """No file of the product package may mention the oracle directory or library."""
    pkg = os.path.join(ROOT, "brotli_g_sdk_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "brotlig_oracle" not in text and "oracle/" not in text, os.path.join(dirpath, f)
