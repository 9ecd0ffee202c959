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
    renames = dict(re.findall(r"^#define\s+(Brotlig\w+)\s+(Brotlig\w+)\s*$", text, flags=re.M))    # symbol-version macros
    names = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text))
    return sorted(renames.get(n, n) for n in names)


def test_library_exports_every_declared_symbol():
    so = ctypes.CDLL(_build.build_hip())
    names = declared_functions()
    assert {"DecompressedSize", "DecodeGPU", "BrotligDecodeBatchDevice_v4", "BrotligDecodeBatchStatus_v4",
            "BrotligDecodeBatchTimed_v4", "BrotligDecodeWorkspaceSize_v4", "BrotligDeviceSelfTest", "BrotligShardPlan",
            "BrotligDecodeBatchMultiDevice_v4", "BrotligContextCreate", "BrotligAbiVersion"} <= set(names)
    for n in names:
        assert hasattr(so, n), n


def test_decompressed_size_needs_no_device():
    from brotli_g_sdk_amd import api
    hdr = np.frombuffer(bytes.fromhex("05fa0300a10f0000"), dtype=np.uint8)      # 3 pages, 64 KiB, last = 1000
    assert api.DecompressedSize(hdr) == 2 * 65536 + 1000


def test_workspace_size_grows_with_streams():
    from brotli_g_sdk_amd import api
    so = api.lib()
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


def test_stale_callers_fail_to_link():
    """The entry points whose contract changed carry the ABI version in their symbol names: the unversioned names a
    caller built against an older header would ask for are not exported."""
    so = ctypes.CDLL(_build.build_hip())
    so.BrotligAbiVersion.restype = ctypes.c_uint32
    assert so.BrotligAbiVersion() == 4
    for n in ("BrotligDecodeBatchDevice", "BrotligDecodeBatchTimed", "BrotligDecodeWorkspaceSize", "BrotligDecodeBatchStatus"):
        assert not hasattr(so, n), n


def test_shard_plan_balances_compressed_bytes():
    """balance by compressed bytes, not stream or page count.  Host arithmetic, no device."""
    from brotli_g_sdk_amd import api
    rng = np.random.default_rng(5)
    for trial in range(200):
        n, g = int(rng.integers(1, 40)), int(rng.integers(1, 10))
        sizes = rng.integers(1, 1000, n) if trial % 3 else rng.integers(1, 10, n) ** 6
        first = api.ShardPlan(sizes, g)
        assert first[0] == 0 and first[-1] == n and all(a <= b for a, b in zip(first, first[1:]))
        runs = [int(sizes[a:b].sum()) for a, b in zip(first, first[1:])]
        assert sum(1 for a, b in zip(first, first[1:]) if b > a) == min(n, g)        # no shard idle while streams remain
        # optimal bottleneck for contiguous runs, by exhaustive dynamic programming
        pre = np.concatenate([[0], np.cumsum(sizes)])
        best = {(0, 0): 0}
        for k in range(1, g + 1):
            for i in range(0, n + 1):
                cands = [max(best[(k - 1, j)], int(pre[i] - pre[j])) for j in range(0, i + 1) if (k - 1, j) in best]
                if cands:
                    best[(k, i)] = min(cands)
        assert max(runs) == best[(g, n)], (sizes, g, first)
    # uneven streams: three small ones weigh as much as the large one
    assert api.ShardPlan([300, 100, 100, 100], 2) == [0, 1, 4]
    assert api.ShardPlan([], 3) == [0, 0, 0, 0]


def test_shard_plan_is_the_same_everywhere_and_even():
    """The plan is host arithmetic: the pure-Python twin (shard.shard_plan, what a rank without hipcc uses) and the exports
    of libbrotlig_hip.so and libbrotlig_cpu.so (one definition, csrc/brotlig_shard_plan.h) agree; equal streams are spread
    evenly (10 streams over 4 ranks used to come out 3,3,3,1)."""
    from brotli_g_sdk_amd import api, cpu, shard
    rng = np.random.default_rng(11)
    for trial in range(300):
        n, g = int(rng.integers(0, 50)), int(rng.integers(1, 12))
        sizes = [1] * n if trial % 4 == 0 else [int(x) for x in (rng.integers(1, 1000, n) if trial % 3 else rng.integers(1, 10, n) ** 6)]
        py = shard.shard_plan(sizes, g)
        assert py == api.ShardPlan(sizes, g) == cpu.ShardPlan(sizes, g), (sizes, g)
        if trial % 4 == 0 and n >= g:
            runs = [b - a for a, b in zip(py, py[1:])]
            assert max(runs) - min(runs) <= 1, (n, g, runs)
    assert [b - a for a, b in zip(*(lambda f: (f, f[1:]))(shard.shard_plan([1] * 10, 4)))] in ([3, 2, 3, 2], [3, 3, 2, 2], [2, 3, 2, 3])
    assert shard.stream_indices(10, 4, 3) == list(range(8, 10))


def test_multi_device_entry_checks_its_arguments_without_a_device():
    from brotli_g_sdk_amd import api
    L = api.lib()
    assert L.BrotligDecodeBatchMultiDevice(None, 1, ctypes.sizeof(api.DeviceBatch), 0, 1, None, None) != 0
    arr = (api.DeviceBatch * 1)()
    assert L.BrotligDecodeBatchMultiDevice(ctypes.addressof(arr), 1, ctypes.sizeof(api.DeviceBatch) - 8, 0, 1, None, None) != 0   # stale struct
    assert L.BrotligDecodeBatchMultiDevice(ctypes.addressof(arr), 1, ctypes.sizeof(api.DeviceBatch), 0, 0, None, None) != 0       # no steps
    assert ctypes.sizeof(api.DeviceBatch) == 104
    for fn in (L.BrotligDecodeBatchMultiDeviceAsync, L.BrotligDecodeBatchMultiDeviceWait):      # the non-blocking pair
        assert fn(None, 1, ctypes.sizeof(api.DeviceBatch)) != 0
        assert fn(ctypes.addressof(arr), 1, ctypes.sizeof(api.DeviceBatch) - 8) != 0
        assert fn(ctypes.addressof(arr), 0, ctypes.sizeof(api.DeviceBatch)) != 0


def test_debug_knobs_are_inert_without_the_environment_switch():
    """BrotligDebugSetDecodeMode / ...Grid are process-wide; they only work in a process started with
    BROTLIG_ENABLE_DEBUG_KNOBS=1 (tests/conftest.py).  A process without it: the library says so, the Python mirror refuses."""
    import subprocess
    import sys
    code = ("import os; os.environ.pop('BROTLIG_ENABLE_DEBUG_KNOBS', None)\n"
            "from brotli_g_sdk_amd import api\n"
            "assert not api.DebugKnobsEnabled()\n"
            "try:\n    api.DebugSetDecodeMode(2)\n    raise SystemExit(3)\nexcept RuntimeError:\n    pass\n"
            "api.DebugSetDecodeMode(0)\n")
    env = {k: v for k, v in os.environ.items() if k != "BROTLIG_ENABLE_DEBUG_KNOBS"}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.run([sys.executable, "-c", code], cwd=root, env=env).returncode == 0
