"""Host-side mirror of the reference's decode interface, over the C ABI in include/brotlig_amd.h.

Same names and argument meaning as the reference (`DecompressedSize`, inc/BrotligDecoder.h:32;
`DecodeGPU`, sample/BrotligGPUDecoder.h:24), plus `BatchDecoder` for the device-pointer batch
entry the benchmark times.  There is no CPU decode path here: if libbrotlig_hip.so is missing or
no HIP device is usable, the calls raise.  torch is used only to own device memory and streams.
"""
import ctypes
import os

import numpy as np

from . import _build

BROTLIG_OK = 0
BROTLIG_ERROR_CORRUPT_STREAM = 14
BROTLIG_ERROR_INCORRECT_STREAM_FORMAT = 15
BROTLIG_ERROR_GENERIC = 16


class BrotligError(RuntimeError):
    def __init__(self, code, what):
        super().__init__(f"{what} failed with BROTLIG_ERROR {code}")
        self.code = code


class _StreamDesc(ctypes.Structure):
    _fields_ = [("in_offset", ctypes.c_uint64), ("out_offset", ctypes.c_uint64),
                ("in_size", ctypes.c_uint64), ("out_capacity", ctypes.c_uint64)]


class DeviceBatch(ctypes.Structure):
    """BrotligDeviceBatch (include/brotlig_amd.h): one shard of a multi-device decode."""
    _fields_ = [("device", ctypes.c_int32), ("num_streams", ctypes.c_uint32),
                ("d_in", ctypes.c_void_p), ("in_bytes", ctypes.c_uint64),
                ("d_out", ctypes.c_void_p), ("out_bytes", ctypes.c_uint64),
                ("d_streams", ctypes.c_void_p),
                ("d_workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_uint64),
                ("d_scratch", ctypes.c_void_p), ("hip_stream", ctypes.c_void_p),
                ("result", ctypes.c_int32), ("reserved", ctypes.c_uint32),
                ("kernel_ms", ctypes.c_double), ("wall_ms", ctypes.c_double)]


ABI_VERSION = 4         # BROTLIG_AMD_ABI_VERSION: the batch entry points carry it in their symbol names
_VERSIONED = ("BrotligDecodeWorkspaceSize", "BrotligDecodeWorkspaceSizeFor", "BrotligDecodeBatchDevice", "BrotligDecodeBatchStatus",
              "BrotligDecodeBatchTimed", "BrotligDecodePhaseProfile", "BrotligDecodeBatchMultiDevice")

_lib = None


class _Lib:
    """The shared library with the header's symbol-version macros applied: L.BrotligDecodeBatchDevice is the symbol
    BrotligDecodeBatchDevice_v4, as it is for a C caller that includes brotlig_amd.h."""

    def __init__(self, cdll, version=ABI_VERSION):
        object.__setattr__(self, "_cdll", cdll)
        object.__setattr__(self, "_version", version)

    def __getattr__(self, name):
        return getattr(self._cdll, f"{name}_v{self._version}" if name in _VERSIONED else name)


def lib():
    """Loads libbrotlig_hip.so (building it in-tree with hipcc if needed).  Raises if absent."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  -- torch's copy of the HIP runtime must be the one in the process: loaded after this library it sees no device
        cdll = ctypes.CDLL(_build.build_hip())
        cdll.BrotligAbiVersion.restype = ctypes.c_uint32
        found = int(cdll.BrotligAbiVersion())
        # only an A/B build of an OLDER source loaded through BROTLIG_HIP_SO (profiles/tools/ab_run.py) may carry another version: the calls
        # this mirror makes have had the same arguments since version 3, and every workspace size is asked of the loaded library
        if found != ABI_VERSION and not (os.environ.get("BROTLIG_HIP_SO") and found == 3):
            raise RuntimeError("libbrotlig_hip.so has ABI version %d, this mirror expects %d" % (found, ABI_VERSION))
        L = _Lib(cdll, found)
        L.BrotligShardPlan.restype = ctypes.c_int
        L.BrotligShardPlan.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
        L.BrotligDecodeBatchMultiDevice.restype = ctypes.c_int
        L.BrotligDecodeBatchMultiDevice.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                                    ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        for name in ("BrotligDecodeBatchMultiDeviceAsync", "BrotligDecodeBatchMultiDeviceWait"):
            getattr(L, name).restype = ctypes.c_int
            getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
        L.BrotligContextCreate.restype = ctypes.c_int
        L.BrotligContextCreate.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.BrotligContextDestroy.restype = None
        L.BrotligContextDestroy.argtypes = [ctypes.c_void_p]
        L.BrotligContextDecodeGPU.restype = ctypes.c_int
        L.BrotligContextDecodeGPU.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32),
                                              ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        L.DecompressedSize.restype = ctypes.c_uint32
        L.DecompressedSize.argtypes = [ctypes.c_void_p]
        L.DecodeGPU.restype = ctypes.c_int
        L.DecodeGPU.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32),
                                ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        L.BrotligDecodeWorkspaceSize.restype = ctypes.c_size_t
        L.BrotligDecodeWorkspaceSize.argtypes = [ctypes.c_uint32]
        L.BrotligDecodeWorkspaceSizeFor.restype = ctypes.c_size_t
        L.BrotligDecodeWorkspaceSizeFor.argtypes = [ctypes.c_uint32, ctypes.c_uint64]
        batch = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint32,
                 ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        L.BrotligDecodeBatchDevice.restype = ctypes.c_int
        L.BrotligDecodeBatchDevice.argtypes = batch
        L.BrotligDecodeBatchStatus.restype = ctypes.c_int
        L.BrotligDecodeBatchStatus.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        try:
            L.BrotligDecodeBatchStreamStatus.restype = ctypes.c_int
            L.BrotligDecodeBatchStreamStatus.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
            L.BrotligStreamerStreamResult.restype = ctypes.c_int
            L.BrotligStreamerStreamResult.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32]
            L.BrotligDebugKnobsEnabled.restype = ctypes.c_uint32
        except AttributeError:
            # only an A/B build of an OLDER source loaded through BROTLIG_HIP_SO (profiles/tools/ab_run.py) may lack the round-5 entries
            if not os.environ.get("BROTLIG_HIP_SO"):
                raise
        L.BrotligDecodeBatchTimed.restype = ctypes.c_int
        L.BrotligDecodeBatchTimed.argtypes = batch + [ctypes.c_uint32, ctypes.c_uint32,
                                                      ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        L.BrotligDecodePhaseProfile.restype = ctypes.c_int
        L.BrotligDecodePhaseProfile.argtypes = batch[:9] + [ctypes.c_void_p, ctypes.c_uint32]
        L.BrotligDeviceSelfTest.restype = ctypes.c_int
        L.BrotligKernelLdsBytes.restype = ctypes.c_uint32
        L.BrotligKernelGridSize.restype = ctypes.c_uint32
        L.BrotligStreamerCreate.restype = ctypes.c_int
        L.BrotligStreamerCreate.argtypes = [ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                            ctypes.POINTER(ctypes.c_void_p)]
        L.BrotligStreamerDestroy.restype = None
        L.BrotligStreamerDestroy.argtypes = [ctypes.c_void_p]
        L.BrotligStreamerSubmit.restype = ctypes.c_int
        L.BrotligStreamerSubmit.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        L.BrotligStreamerWait.restype = ctypes.c_int
        L.BrotligStreamerWait.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        L.BrotligStreamerOutput.restype = ctypes.c_void_p
        L.BrotligStreamerOutput.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        try:
            L.BrotligStreamerCreateDeviceOutput.restype = ctypes.c_int
            L.BrotligStreamerCreateDeviceOutput.argtypes = L.BrotligStreamerCreate.argtypes
            L.BrotligStreamerDeviceOutput.restype = ctypes.c_int
            L.BrotligStreamerDeviceOutput.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p),
                                                      ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_void_p)]
            L.BrotligStreamerConsumerDone.restype = ctypes.c_int
            L.BrotligStreamerConsumerDone.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
            L.BrotligStreamerStreamWait.restype = ctypes.c_int
            L.BrotligStreamerStreamWait.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
            L.BrotligStreamerAcquire.restype = ctypes.c_int
            L.BrotligStreamerAcquire.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
            L.BrotligStreamerSubmitInPlace.restype = ctypes.c_int
            L.BrotligStreamerSubmitInPlace.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                       ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        except AttributeError:
            if not os.environ.get("BROTLIG_HIP_SO"):        # (an A/B build of an older source)
                raise
        _lib = L
    return _lib


def DecompressedSize(src) -> int:
    """uint32_t DecompressedSize(uint8_t* src) -- inc/BrotligDecoder.h:32."""
    a = np.ascontiguousarray(np.frombuffer(src, dtype=np.uint8) if not isinstance(src, np.ndarray) else src, dtype=np.uint8)
    return int(lib().DecompressedSize(a.ctypes.data))


def DecodeGPU(src, output_size=None):
    """BROTLIG_ERROR DecodeGPU(useWarpDevice, input_size, input, output_size, output, time)
    -- sample/BrotligGPUDecoder.h:24.  Host bytes in, host bytes out.
    Returns (output ndarray, kernel_time_ms)."""
    a = np.ascontiguousarray(np.frombuffer(src, dtype=np.uint8) if not isinstance(src, np.ndarray) else src, dtype=np.uint8)
    cap = DecompressedSize(a) if output_size is None else int(output_size)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    osz = ctypes.c_uint32(cap)
    t = ctypes.c_double(0.0)
    rc = lib().DecodeGPU(0, len(a), a.ctypes.data, ctypes.byref(osz), out.ctypes.data, ctypes.byref(t))
    if rc != BROTLIG_OK:
        raise BrotligError(rc, "DecodeGPU")
    return out[:osz.value], t.value


class Context:
    """BrotligContext: DecodeGPU with device buffers, stream and events kept between calls."""

    def __init__(self, device=-1):
        self._h = ctypes.c_void_p()
        rc = lib().BrotligContextCreate(int(device), ctypes.byref(self._h))
        if rc != BROTLIG_OK:
            raise BrotligError(rc, "BrotligContextCreate")

    def close(self):
        if self._h:
            lib().BrotligContextDestroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def DecodeGPU(self, src, output_size=None):
        a = np.ascontiguousarray(np.frombuffer(src, dtype=np.uint8) if not isinstance(src, np.ndarray) else src, dtype=np.uint8)
        cap = DecompressedSize(a) if output_size is None else int(output_size)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        osz = ctypes.c_uint32(cap)
        t = ctypes.c_double(0.0)
        rc = lib().BrotligContextDecodeGPU(self._h, len(a), a.ctypes.data, ctypes.byref(osz), out.ctypes.data, ctypes.byref(t))
        if rc != BROTLIG_OK:
            raise BrotligError(rc, "BrotligContextDecodeGPU")
        return out[:osz.value], t.value


def ShardPlan(in_sizes, num_shards):
    """BrotligShardPlan: contiguous runs of streams, one per shard, balanced by compressed bytes.  Returns the
    num_shards + 1 run boundaries.  Host arithmetic only (no device needed)."""
    sizes = np.ascontiguousarray(in_sizes, dtype=np.uint64)
    first = np.zeros(int(num_shards) + 1, dtype=np.uint32)
    rc = lib().BrotligShardPlan(sizes.ctypes.data, len(sizes), int(num_shards), first.ctypes.data)
    if rc != BROTLIG_OK:
        raise BrotligError(rc, "BrotligShardPlan")
    return [int(x) for x in first]


def _device_batches(decoders, streams):
    n = len(decoders)
    arr = (DeviceBatch * n)()
    for i, d in enumerate(decoders):
        b = arr[i]
        b.device = d.device.index if d.device.index is not None else d.torch.cuda.current_device()
        b.num_streams = d.n
        b.d_in, b.in_bytes = d.d_in.data_ptr(), d.in_bytes
        b.d_out, b.out_bytes = d.d_out.data_ptr(), d.out_bytes
        b.d_streams = d.d_desc.data_ptr()
        b.d_workspace, b.workspace_bytes = d.d_ws.data_ptr(), d.ws_bytes
        b.d_scratch = d.d_scratch.data_ptr() if d.d_scratch is not None else None
        b.hip_stream = streams[i] if streams is not None else None
    return arr


def DecodeBatchMultiDevice(decoders, warmup=0, steps=1, streams=None):
    """BrotligDecodeBatchMultiDevice over `decoders` (BatchDecoder objects, one per shard, each on its own device -- or
    several on one).  Returns (max kernel ms, max wall ms, per-shard list of (result, kernel_ms, wall_ms))."""
    n = len(decoders)
    arr = _device_batches(decoders, streams)
    mk, mw = ctypes.c_double(0.0), ctypes.c_double(0.0)
    rc = lib().BrotligDecodeBatchMultiDevice(ctypes.addressof(arr), n, ctypes.sizeof(DeviceBatch), int(warmup), int(steps),
                                             ctypes.byref(mk), ctypes.byref(mw))
    per = [(arr[i].result, arr[i].kernel_ms, arr[i].wall_ms) for i in range(n)]
    if rc != BROTLIG_OK:
        raise BrotligError(rc, "BrotligDecodeBatchMultiDevice")
    return mk.value, mw.value, per


class MultiDeviceAsync:
    """The non-blocking pair BrotligDecodeBatchMultiDeviceAsync / ...Wait: enqueue every shard on its device and stream and
    return; wait() collects the batch status of each shard.  No host thread is created, nothing is timed."""

    def __init__(self, decoders, streams=None):
        self.n = len(decoders)
        self.arr = _device_batches(decoders, streams)
        self.keep = (decoders, streams)
        rc = lib().BrotligDecodeBatchMultiDeviceAsync(ctypes.addressof(self.arr), self.n, ctypes.sizeof(DeviceBatch))
        if rc != BROTLIG_OK:
            # the C side has enqueued every shard it could (it records a result per shard and goes on): their kernels are writing the
            # decoders' buffers, so they are waited for before the half-built object is dropped -- Wait only touches the
            # shards whose enqueue succeeded
            lib().BrotligDecodeBatchMultiDeviceWait(ctypes.addressof(self.arr), self.n, ctypes.sizeof(DeviceBatch))
            raise BrotligError(rc, "BrotligDecodeBatchMultiDeviceAsync")

    def wait(self):
        rc = lib().BrotligDecodeBatchMultiDeviceWait(ctypes.addressof(self.arr), self.n, ctypes.sizeof(DeviceBatch))
        results = [self.arr[i].result for i in range(self.n)]
        if rc != BROTLIG_OK:
            raise BrotligError(rc, "BrotligDecodeBatchMultiDeviceWait")
        return results


def _debug_knobs(L):
    """The two diagnostics switches are inert unless the process has BROTLIG_ENABLE_DEBUG_KNOBS=1 in its environment when the library
    first launches (the library reads it once); tests/conftest.py sets it.  Asking for a knob in a process without it is an error here
    rather than a silent no-op."""
    if os.environ.get("BROTLIG_ENABLE_DEBUG_KNOBS") != "1" or not L.BrotligDebugKnobsEnabled():
        raise RuntimeError("BrotligDebugSet* are inert: start the process with BROTLIG_ENABLE_DEBUG_KNOBS=1 (diagnostics only)")


def DebugKnobsEnabled():
    return bool(lib().BrotligDebugKnobsEnabled())


def DebugSetDecodeGrid(workgroups):
    """BrotligDebugSetDecodeGrid (diagnostics): 0 = the normal launch rule."""
    L = lib()
    if workgroups:
        _debug_knobs(L)
    L.BrotligDebugSetDecodeGrid.restype = None
    L.BrotligDebugSetDecodeGrid.argtypes = [ctypes.c_uint32]
    L.BrotligDebugSetDecodeGrid(int(workgroups))


def DebugSetDecodeMode(mode):
    """BrotligDebugSetDecodeMode (diagnostics): 0 = the normal rule, 1 = never two wavefronts per page, 2 = always."""
    L = lib()
    if mode:
        _debug_knobs(L)
    L.BrotligDebugSetDecodeMode.restype = None
    L.BrotligDebugSetDecodeMode.argtypes = [ctypes.c_uint32]
    L.BrotligDebugSetDecodeMode(int(mode))


def DeviceSelfTest():
    rc = lib().BrotligDeviceSelfTest()
    if rc != BROTLIG_OK:
        raise BrotligError(rc, "BrotligDeviceSelfTest")


class BatchDecoder:
    """Owns the device buffers for a batch of streams and decodes them with one enqueue
    (BrotligDecodeBatchDevice).  `streams` is a list of uint8 arrays, each one .brotlig stream.
    `schedule=False` sizes the workspace at its minimum, which turns the page schedule off (pages are
    then decoded in stream order)."""

    def __init__(self, streams, device="cuda:0", out_sizes=None, schedule=True):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        L = lib()
        n = len(streams)
        sizes = [int(DecompressedSize(s)) for s in streams] if out_sizes is None else [int(x) for x in out_sizes]
        in_offs, pos = [], 0
        for s in streams:
            in_offs.append(pos)
            pos += (len(s) + 15) // 16 * 16
        self.in_bytes = pos
        out_offs, opos = [], 0
        precon = False
        for s, sz in zip(streams, sizes):
            out_offs.append(opos)
            page = 32768 << (int(s[4]) & 3)
            precon = precon or bool((int(s[6]) >> 4) & 1)
            opos += (sz + page - 1) // page * page
        self.out_bytes = opos
        self.sizes, self.out_offs, self.n = sizes, out_offs, n
        self.compressed_bytes = int(sum(len(s) for s in streams))
        self.decompressed_bytes = int(sum(sizes))
        host_in = np.zeros(pos + 64, dtype=np.uint8)
        for o, s in zip(in_offs, streams):
            host_in[o:o + len(s)] = s
        desc = np.zeros((n, 4), dtype=np.uint64)                    # BrotligStreamDesc
        desc[:, 0] = in_offs
        desc[:, 1] = out_offs
        desc[:, 2] = [len(s) for s in streams]
        desc[:, 3] = [(out_offs[i + 1] if i + 1 < n else opos) - out_offs[i] for i in range(n)]
        self.d_in = torch.from_numpy(host_in).to(self.device)
        self.d_desc = torch.from_numpy(desc.view(np.int64)).to(self.device)
        self.d_out = torch.empty(opos + 64, dtype=torch.uint8, device=self.device)
        self.d_scratch = torch.empty(opos + 64, dtype=torch.uint8, device=self.device) if precon else None
        self.ws_bytes = int(L.BrotligDecodeWorkspaceSizeFor(n, opos) if schedule else L.BrotligDecodeWorkspaceSize(n))
        self.d_ws = torch.zeros(self.ws_bytes, dtype=torch.uint8, device=self.device)

    def _args(self, stream):
        scratch = self.d_scratch.data_ptr() if self.d_scratch is not None else None
        return [self.d_in.data_ptr(), self.in_bytes, self.d_out.data_ptr(), self.out_bytes, self.d_desc.data_ptr(),
                self.n, self.d_ws.data_ptr(), self.ws_bytes, scratch, stream]

    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def decode(self, check=True):
        """Enqueue one decode of the whole batch on torch's current stream."""
        with self.torch.cuda.device(self.device):
            st = self._stream()
            rc = lib().BrotligDecodeBatchDevice(*self._args(st))
            if rc != BROTLIG_OK:
                raise BrotligError(rc, "BrotligDecodeBatchDevice")
            if check:
                rc = lib().BrotligDecodeBatchStatus(self.d_ws.data_ptr(), st)
                if rc != BROTLIG_OK:
                    raise BrotligError(rc, "BrotligDecodeBatchStatus")

    def timed(self, warmup, steps, check=True):
        """Returns (total_ms over `steps` passes, average decode-kernel ms), both from HIP events on
        the launch stream.  check=False leaves the batch status to the caller (status(): a copy and a wait that a caller timing the
        steps with its own clock does after its clock has stopped)."""
        with self.torch.cuda.device(self.device):
            st = self._stream()
            total, kern = ctypes.c_double(0.0), ctypes.c_double(0.0)
            rc = lib().BrotligDecodeBatchTimed(*self._args(st), warmup, steps, ctypes.byref(total), ctypes.byref(kern))
            if rc != BROTLIG_OK:
                raise BrotligError(rc, "BrotligDecodeBatchTimed")
            if check:
                self.status()
            return total.value, kern.value

    def status(self):
        """BrotligDecodeBatchStatus of the last decode (waits for the stream); raises on a failed batch."""
        with self.torch.cuda.device(self.device):
            rc = lib().BrotligDecodeBatchStatus(self.d_ws.data_ptr(), self._stream())
            if rc != BROTLIG_OK:
                raise BrotligError(rc, "BrotligDecodeBatchStatus")

    def stream_status(self):
        """BrotligDecodeBatchStreamStatus: (batch result, [BROTLIG_ERROR per stream]) of the last decode -- which assets were damaged."""
        with self.torch.cuda.device(self.device):
            res = np.zeros(self.n, dtype=np.int32)
            rc = lib().BrotligDecodeBatchStreamStatus(self.d_ws.data_ptr(), self.n, res.ctypes.data, self._stream())
            return int(rc), [int(x) for x in res]

    PHASES = ("setup", "tables", "commands", "ring", "positions", "literals", "group_setup", "level_tail",
              "delta", "total", "rounds", "levels", "lv_short", "lv_bytes", "lv_long_and_far", "solo_rounds",
              "cmd_symbol", "cmd_extra_bits", "slide", "pieces_and_far_loads", "bitmaps", "groups", "lit_steps", "lv_overlap", "team_levels",
              "level_halves", "group_halves")

    def phase_profile(self):
        """Per-phase shader-clock sums from the phase-timer twin of the decode kernel (diagnostics)."""
        with self.torch.cuda.device(self.device):
            self.torch.cuda.synchronize()
            out = np.zeros(len(self.PHASES), dtype=np.uint64)
            rc = lib().BrotligDecodePhaseProfile(*self._args(None)[:9], out.ctypes.data, len(out))
            if rc != BROTLIG_OK:
                raise BrotligError(rc, "BrotligDecodePhaseProfile")
            return dict(zip(self.PHASES, (int(x) for x in out)))

    def wave_times(self):
        """(first, last) tick of a 100 MHz counter for every wavefront of one launch of the phase-timer twin (diagnostics: the tail)."""
        with self.torch.cuda.device(self.device):
            self.torch.cuda.synchronize()
            grid = int(lib().BrotligKernelGridSize())
            out = np.zeros(len(self.PHASES) + 2 * grid, dtype=np.uint64)
            rc = lib().BrotligDecodePhaseProfile(*self._args(None)[:9], out.ctypes.data, len(out))
            if rc != BROTLIG_OK:
                raise BrotligError(rc, "BrotligDecodePhaseProfile")
            return out[len(self.PHASES):].reshape(grid, 2)

    def schedule_times(self, tickets=4096):
        """(came, ticket, phase open, done) ticks of a 100 MHz counter for the first `tickets` workgroups -- by ticket -- of the schedule kernel of
        one launch (diagnostics, round 6: where the one kernel in front of the page decode spends its time)."""
        with self.torch.cuda.device(self.device):
            self.torch.cuda.synchronize()
            grid = int(lib().BrotligKernelGridSize())
            out = np.zeros(len(self.PHASES) + 2 * grid + 4 * tickets, dtype=np.uint64)
            rc = lib().BrotligDecodePhaseProfile(*self._args(None)[:9], out.ctypes.data, len(out))
            if rc != BROTLIG_OK:
                raise BrotligError(rc, "BrotligDecodePhaseProfile")
            return out[len(self.PHASES) + 2 * grid:].reshape(tickets, 4)

    def output(self, i):
        """Decompressed bytes of stream i as a host uint8 array."""
        o = self.out_offs[i]
        return self.d_out[o:o + self.sizes[i]].cpu().numpy()

    def poison_output(self, value=0xCD):
        self.d_out.fill_(value)


class Streamer:
    """Asynchronous host-to-host decoding of batches of streams (BrotligStreamer*, include/brotlig_amd.h):
    a ring of slots with pinned staging, one hipStream_t each, so that the upload of a batch overlaps
    the decode of the previous one and the download of the one before.

        st = Streamer(slots=3, slot_in_bytes=64 << 20, slot_out_bytes=256 << 20)
        t = st.submit(streams)                  # returns at once
        outs = st.result(t)                     # list of uint8 arrays (copies out of the pinned buffer)
    """

    def __init__(self, slots=3, slot_in_bytes=64 << 20, slot_out_bytes=256 << 20, max_streams=4096, device_output=False):
        """device_output=True (round 6): the decoded bytes stay in device memory -- nothing is downloaded, no pinned staging for them;
        `device_output(ticket, i)` says where stream i of a batch is and which event to wait for."""
        self._h = ctypes.c_void_p()
        self.device_output_mode = bool(device_output)
        create = lib().BrotligStreamerCreateDeviceOutput if device_output else lib().BrotligStreamerCreate
        rc = create(slots, slot_in_bytes, slot_out_bytes, max_streams, ctypes.byref(self._h))
        if rc != BROTLIG_OK:
            raise BrotligError(rc, "BrotligStreamerCreateDeviceOutput" if device_output else "BrotligStreamerCreate")
        self._keep = {}

    def close(self):
        if self._h:
            lib().BrotligStreamerDestroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, streams, outputs=None):
        """streams: list of uint8 arrays / bytes.  outputs: optional list of writable uint8 arrays (or None
        entries) that receive the decoded bytes at wait(); without it the bytes stay in pinned memory."""
        arrs = [np.ascontiguousarray(np.frombuffer(s, dtype=np.uint8) if not isinstance(s, np.ndarray) else s, dtype=np.uint8)
                for s in streams]
        n = len(arrs)
        ins = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrs])
        sizes = (ctypes.c_uint32 * n)(*[a.size for a in arrs])
        outs = caps = None
        if outputs is not None:
            outs = (ctypes.c_void_p * n)(*[(o.ctypes.data if o is not None else None) for o in outputs])
            caps = (ctypes.c_uint32 * n)(*[(o.size if o is not None else 0) for o in outputs])
        t = ctypes.c_uint64()
        rc = lib().BrotligStreamerSubmit(self._h, n, ins, sizes, outs, caps, ctypes.byref(t))
        if rc != BROTLIG_OK:
            raise BrotligError(rc, "BrotligStreamerSubmit")
        self._keep[t.value] = (n, outputs)          # the caller's output arrays must outlive the batch
        return t.value

    def acquire(self):
        """BrotligStreamerAcquire: the pinned staging area of the slot the next batch goes to, as a writable uint8 array (no copy) -- a loader
        reads its streams straight into it, at 16-byte aligned ascending offsets, and calls submit_in_place."""
        p, cap = ctypes.c_void_p(), ctypes.c_uint64()
        rc = lib().BrotligStreamerAcquire(self._h, ctypes.byref(p), ctypes.byref(cap))
        if rc != BROTLIG_OK:
            raise BrotligError(rc, "BrotligStreamerAcquire")
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(cap.value,))

    def submit_in_place(self, offsets, sizes, outputs=None):
        """BrotligStreamerSubmitInPlace: the streams lie in the acquired staging area at `offsets` (bytes `sizes`)."""
        n = len(offsets)
        offs = (ctypes.c_uint64 * n)(*[int(o) for o in offsets])
        szs = (ctypes.c_uint32 * n)(*[int(x) for x in sizes])
        outs = caps = None
        if outputs is not None:
            outs = (ctypes.c_void_p * n)(*[(o.ctypes.data if o is not None else None) for o in outputs])
            caps = (ctypes.c_uint32 * n)(*[(o.size if o is not None else 0) for o in outputs])
        t = ctypes.c_uint64()
        rc = lib().BrotligStreamerSubmitInPlace(self._h, n, offs, szs, outs, caps, ctypes.byref(t))
        if rc != BROTLIG_OK:
            raise BrotligError(rc, "BrotligStreamerSubmitInPlace")
        self._keep[t.value] = (n, outputs)
        return t.value

    def wait(self, ticket):
        """Waits for the batch; its `outputs` arrays (if any were given) are filled on return.  Also valid for a
        batch that a later submit() had to complete to make room (one generation back)."""
        rc = lib().BrotligStreamerWait(self._h, ticket)
        outputs = self._keep.get(ticket, (0, None))[1]
        if outputs is not None or rc != BROTLIG_OK:
            self._keep.pop(ticket, None)            # nothing left to fetch for this ticket
        if rc != BROTLIG_OK:
            raise BrotligError(rc, "BrotligStreamerWait")

    def stream_results(self, ticket, n):
        """BrotligStreamerStreamResult for streams 0..n-1 of the batch: BROTLIG_ERROR per stream (waits for the batch)."""
        return [int(lib().BrotligStreamerStreamResult(self._h, ticket, i)) for i in range(n)]

    def output(self, ticket, index):
        """BrotligStreamerOutput: the decoded bytes of one stream (a copy), or None for a damaged stream / an unknown ticket."""
        sz = ctypes.c_uint32()
        p = lib().BrotligStreamerOutput(self._h, ticket, index, ctypes.byref(sz))
        if not p:
            return None
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(sz.value,)).copy()

    def device_output(self, ticket, index):
        """BrotligStreamerDeviceOutput: (device pointer, size, hipEvent_t) of one stream of a batch of a device-output streamer.  Does not wait:
        a consumer makes its stream wait for the event.  Valid until the slot's next batch is submitted."""
        p, sz, ev = ctypes.c_void_p(), ctypes.c_uint32(), ctypes.c_void_p()
        rc = lib().BrotligStreamerDeviceOutput(self._h, ticket, index, ctypes.byref(p), ctypes.byref(sz), ctypes.byref(ev))
        if rc != BROTLIG_OK:
            raise BrotligError(rc, "BrotligStreamerDeviceOutput")
        return p.value, sz.value, ev.value

    def device_tensor(self, ticket, index, torch_stream=None):
        """The same as a torch uint8 tensor over the streamer's device memory (no copy); torch's current stream (or `torch_stream`) is made to
        wait for the batch's event first, so that work enqueued on it afterwards sees the decoded bytes."""
        import torch
        p, sz, ev = self.device_output(ticket, index)
        s = torch_stream if torch_stream is not None else torch.cuda.current_stream()
        rc = lib().BrotligStreamerStreamWait(self._h, ticket, ctypes.c_void_p(s.cuda_stream))
        if rc != BROTLIG_OK:
            raise BrotligError(rc, "BrotligStreamerStreamWait")

        class _Mem:                                     # __cuda_array_interface__: torch wraps foreign device memory without copying
            __cuda_array_interface__ = {"shape": (sz,), "typestr": "|u1", "data": (p, False), "version": 2}
        return torch.as_tensor(_Mem(), device="cuda") if sz else torch.empty(0, dtype=torch.uint8, device="cuda")

    def consumer_done(self, ticket, torch_stream=None):
        """BrotligStreamerConsumerDone: everything enqueued so far on torch's current stream (or `torch_stream`) is what reads the batch; the
        batch that reuses its slot waits for it on the device."""
        import torch
        s = torch_stream if torch_stream is not None else torch.cuda.current_stream()
        rc = lib().BrotligStreamerConsumerDone(self._h, ticket, ctypes.c_void_p(s.cuda_stream))
        if rc != BROTLIG_OK:
            raise BrotligError(rc, "BrotligStreamerConsumerDone")

    def result(self, ticket):
        """Waits for the batch and returns its decoded streams as fresh arrays."""
        n, outputs = self._keep[ticket]
        self.wait(ticket)
        if outputs is not None:                     # the caller's own arrays hold the bytes
            return [o for o in outputs]
        self._keep.pop(ticket, None)
        out = []
        for i in range(n):
            sz = ctypes.c_uint32()
            p = lib().BrotligStreamerOutput(self._h, ticket, i, ctypes.byref(sz))
            if not p:
                raise BrotligError(BROTLIG_ERROR_GENERIC, "BrotligStreamerOutput")
            out.append(np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(sz.value,)).copy())
        return out
