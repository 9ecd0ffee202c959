"""tools/brotlig: the portable command-line tool (switches of sample/brotlig_cli.cpp:174-327)."""
import os
import subprocess

import numpy as np
import pytest

from brotli_g_sdk_amd import _build
from brotli_g_sdk_amd import datagen as D
from helpers import oracle_decode


@pytest.fixture(scope="module")
def cli():
    _build.build_encoder()
    _build.build_hip()
    return _build.build_cli()


def test_cli_compress_matches_oracle(cli, tmp_path):
    """`brotlig file` writes file.brotlig; the oracle decodes it back to the input (CPU only)."""
    data = D.mixed(65536 * 3 + 777, 9)
    src = tmp_path / "asset.bin"
    data.tofile(src)
    r = subprocess.run([cli, "-pagesize", "32768", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    stream = np.fromfile(str(src) + ".brotlig", dtype=np.uint8)
    rc, out = oracle_decode(stream)
    assert rc == 0 and np.array_equal(out, data)
    # pre-conditioning switches: pixels in, 4x4 blocks in the header
    tex = D.bc_texture(1, 64, 32, seed=4)
    tsrc = tmp_path / "tex.bin"
    tex.tofile(tsrc)
    r = subprocess.run([cli, "-precondition", "-swizzle", "-delta-encode", "-data-format", "1", "-texture-width", "256",
                        "-texture-height", "128", str(tsrc), str(tmp_path / "tex.out.brotlig")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rc, out = oracle_decode(np.fromfile(tmp_path / "tex.out.brotlig", dtype=np.uint8), out_size=len(tex))
    assert rc == 0 and np.array_equal(out, tex)


def test_cli_rejects_bad_usage(cli, tmp_path):
    assert subprocess.run([cli], capture_output=True).returncode == 1
    assert subprocess.run([cli, "-no-such-switch", "x"], capture_output=True).returncode == 1
    assert subprocess.run([cli, str(tmp_path / "missing.bin")], capture_output=True).returncode == 2
    src = tmp_path / "t.bin"
    D.text(1000, 1).tofile(src)
    assert subprocess.run([cli, "-precondition", str(src)], capture_output=True).returncode == 2      # format / size missing


def test_cli_cpu_round_trip(cli, tmp_path):
    """`-cpu` decompresses with DecodeCPU (libbrotlig_cpu.so): an explicit choice, never a fallback."""
    for name, data, extra in (("a.bin", D.mixed(65536 * 5 + 123, 3), []),
                              ("t.bin", D.bc_texture(3, 64, 64, seed=2), ["-precondition", "-swizzle", "-delta-encode", "-data-format", "3",
                                                                         "-texture-width", "256", "-texture-height", "256"])):
        src = tmp_path / name
        data.tofile(src)
        assert subprocess.run([cli] + extra + [str(src)], capture_output=True).returncode == 0
        os.rename(src, str(src) + ".orig")
        r = subprocess.run([cli, "-cpu", "-num-repeat", "2", str(src) + ".brotlig"], capture_output=True, text=True)
        assert r.returncode == 0 and "on the CPU" in r.stdout, r.stderr
        assert np.array_equal(np.fromfile(src, dtype=np.uint8), data)


@pytest.mark.gpu
def test_cli_round_trip_on_gpu(cli, tmp_path):
    """compress, then `brotlig file.brotlig` decompresses on the GPU to the original bytes."""
    for name, data, extra in (("a.bin", D.mixed(65536 * 5 + 123, 3), []),
                              ("t.bin", D.bc_texture(3, 64, 64, seed=2), ["-precondition", "-swizzle", "-delta-encode", "-data-format", "3",
                                                                         "-texture-width", "256", "-texture-height", "256"])):
        src = tmp_path / name
        data.tofile(src)
        assert subprocess.run([cli] + extra + [str(src)], capture_output=True).returncode == 0
        os.rename(src, str(src) + ".orig")
        r = subprocess.run([cli, "-gpu", "-num-repeat", "2", str(src) + ".brotlig"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "GB/s decompressed" in r.stdout
        assert np.array_equal(np.fromfile(src, dtype=np.uint8), data)
