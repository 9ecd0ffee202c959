"""bench.py attaches the committed HBM-traffic measurement to its line only when the library it benchmarks holds the SAME decode kernel the
measurement was taken on -- by the hash of the kernel's gfx950 disassembly (bench.kernel_disasm_hash), not by a hand-kept list of
"equivalent" sources.  hipcc cross-compiles here; no GPU needed."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

from helpers import ROOT

sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "brotli_g_sdk_amd", "csrc")


def _variant(path, flags):
    from brotli_g_sdk_amd._build import HIP_FLAGS
    subprocess.check_call(["hipcc"] + HIP_FLAGS + ["-fPIC", "-shared", "-I", os.path.join(ROOT, "include"),
                           "-I", CSRC, "-o", path, os.path.join(CSRC, "brotlig_hip.hip")] + flags, cwd=CSRC,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return path


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc (cross-compiles gfx950 without a GPU)")
def test_disasm_hash_follows_the_decode_kernel_and_nothing_else(tmp_path):
    import bench
    from brotli_g_sdk_amd import _build
    base = bench.kernel_disasm_hash(_build.build_hip())
    assert base is not None and len(base) == 16
    with ThreadPoolExecutor(2) as ex:
        other = ex.submit(_variant, str(tmp_path / "dc.so"), ["-DBROTLIG_TUNE_DC_ASM_UNROLL=2"])          # another de-conditioning kernel: the decode kernel moves, its code does not change
        changed = ex.submit(_variant, str(tmp_path / "chunks.so"), ["-DBROTLIG_TUNE_CHUNKS=2"])           # another decode kernel
        assert bench.kernel_disasm_hash(other.result()) == base
        assert bench.kernel_disasm_hash(changed.result()) not in (None, base)
    assert bench.kernel_disasm_hash(str(tmp_path / "missing.so")) is None


def test_committed_traffic_files_carry_no_self_granted_equivalences():
    """The newest traffic file is keyed by the disassembly hash; none of this round's files carries an `also_valid_for` list."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[5-9]_*hbm_traffic.json")))
    for f in files:
        t = json.load(open(f))
        assert "also_valid_for" not in t and len(t.get("kernel_disasm_sha16") or "") == 16, f
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "also_valid_for" not in src
