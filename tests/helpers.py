"""Shared test helpers: ctypes bindings for the oracle (checker) and the encoder (input
generator), and seeded synthetic data classes matching BASELINE.md section 5."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "brotli_g_sdk_amd", "csrc")
ORACLE_DIR = os.path.join(ROOT, "oracle")


def _build(path, cmd, cwd):
    if not os.path.exists(path):
        subprocess.check_call(cmd, cwd=cwd)


def oracle_lib():
    so = os.path.join(ORACLE_DIR, "libbrotlig_oracle.so")
    _build(so, ["make", "-s"], ORACLE_DIR)
    L = ctypes.CDLL(so)
    L.DecompressedSize.restype = ctypes.c_uint32
    L.DecompressedSize.argtypes = [ctypes.c_void_p]
    L.DecodeCPU.restype = ctypes.c_int
    L.DecodeCPU.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p, ctypes.c_void_p]
    L.brotlig_oracle_decode.restype = ctypes.c_int
    L.brotlig_oracle_decode.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p,
                                        ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.brotlig_oracle_cmd_lut.restype = None
    L.brotlig_oracle_cmd_lut.argtypes = [ctypes.c_uint32] + [ctypes.POINTER(ctypes.c_uint32)] * 4 + [ctypes.POINTER(ctypes.c_int)]
    L.brotlig_oracle_decondition_addr.restype = ctypes.c_uint32
    L.brotlig_oracle_decondition_addr.argtypes = [ctypes.c_uint32] * 4
    return L


def oracle_decode(stream: np.ndarray, workers: int = 0, out_size: int = None):
    """Returns (rc, output ndarray).  `stream` is a uint8 array holding one .brotlig stream."""
    L = oracle_lib()
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    n = L.DecompressedSize(stream.ctypes.data) if out_size is None else out_size
    out = np.full(n + 64, 0xA5, dtype=np.uint8)
    osz = ctypes.c_uint32(n)
    rc = L.brotlig_oracle_decode(len(stream), stream.ctypes.data, ctypes.byref(osz), out.ctypes.data, workers, None)
    assert np.all(out[n:] == 0xA5), "oracle wrote past the output"
    return rc, out[:osz.value].copy()
