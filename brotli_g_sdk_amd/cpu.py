"""Host-side mirror of the reference's CPU decode entry (`DecodeCPU`, inc/BrotligDecoder.h:33) over
libbrotlig_cpu.so (include/brotlig_amd_cpu.h).  Deliberately not part of `api`: the GPU path never imports this
module and has no CPU fallback."""
import ctypes

import numpy as np

from . import _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(_build.build_cpu())
        L.DecompressedSize.restype = ctypes.c_uint32
        L.DecompressedSize.argtypes = [ctypes.c_void_p]
        L.DecodeCPU.restype = ctypes.c_int
        L.DecodeCPU.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p, ctypes.c_void_p]
        L.BrotligDecodeCPU.restype = ctypes.c_int
        L.BrotligDecodeCPU.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p, ctypes.c_uint32]
        L.BrotligDecodeCPUWithFeedback.restype = ctypes.c_int
        L.BrotligDecodeCPUWithFeedback.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p,
                                                   ctypes.c_uint32, FEEDBACK_PROC, ctypes.c_void_p]
        L.BrotligShardPlan.restype = ctypes.c_int
        L.BrotligShardPlan.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
        _lib = L
    return _lib


def ShardPlan(in_sizes, num_shards):
    """BrotligShardPlan as exported by libbrotlig_cpu.so (the same code as libbrotlig_hip.so's: csrc/brotlig_shard_plan.h)."""
    sizes = np.ascontiguousarray(in_sizes, dtype=np.uint64)
    first = np.zeros(int(num_shards) + 1, dtype=np.uint32)
    rc = lib().BrotligShardPlan(sizes.ctypes.data, len(sizes), int(num_shards), first.ctypes.data)
    if rc != 0:
        raise ValueError(f"BrotligShardPlan failed with {rc}")
    return [int(x) for x in first]


# int (*BrotligFeedbackProc)(int type, const char* message, void* user) -- include/brotlig_amd_cpu.h
FEEDBACK_PROC = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p)
BROTLIG_PROGRESS, BROTLIG_WARNING = 0, 1
BROTLIG_ABORTED = 1


def DecodeCPU(src, output_size=None, workers=None, feedbackProc=None):
    """BROTLIG_ERROR DecodeCPU(input_size, src, output_size, output, feedbackProc).  Returns (code, output ndarray).
    `feedbackProc(type, message) -> bool` is the reference's callback (inc/common/BrotligCommon.h:92): called once per
    page from the decoding threads, True aborts (code BROTLIG_ABORTED).  It travels through the C twin
    BrotligDecodeCPUWithFeedback; the reference-named entry itself only accepts NULL."""
    a = np.ascontiguousarray(np.frombuffer(src, dtype=np.uint8) if not isinstance(src, np.ndarray) else src, dtype=np.uint8)
    cap = int(lib().DecompressedSize(a.ctypes.data)) if output_size is None and len(a) >= 8 else int(output_size or 0)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    osz = ctypes.c_uint32(cap)
    if feedbackProc is not None:
        cb = FEEDBACK_PROC(lambda t, m, u: 1 if feedbackProc(t, m.decode()) else 0)
        rc = lib().BrotligDecodeCPUWithFeedback(len(a), a.ctypes.data, ctypes.byref(osz), out.ctypes.data, int(workers or 0), cb, None)
    elif workers is None:
        rc = lib().DecodeCPU(len(a), a.ctypes.data, ctypes.byref(osz), out.ctypes.data, None)
    else:
        rc = lib().BrotligDecodeCPU(len(a), a.ctypes.data, ctypes.byref(osz), out.ctypes.data, int(workers))
    return rc, out[:osz.value] if rc == 0 else out[:0]
