"""The reference-side bindings INTEGRATION.md shows are compiled and linked here (g++ against libbrotlig_hip.so /
libbrotlig_cpu.so), so that the documented drop-in is known to build: the C++ shim for the sample's DecodeGPU
(sample/BrotligGPUDecoder.h:24), a plain-C translation unit that includes both public headers, and a C caller of
the multi-device entry.  Nothing is executed on a device (CPU test); the -m gpu twin runs the multi-device program."""
import os
import re
import shutil
import subprocess

import pytest

from brotli_g_sdk_amd import _build
from helpers import ROOT

INC = os.path.join(ROOT, "include")
CSRC = os.path.join(ROOT, "brotli_g_sdk_amd", "csrc")
DOC = os.path.join(ROOT, "INTEGRATION.md")


def _doc_block(marker):
    """The fenced code block of INTEGRATION.md whose first line contains `marker`."""
    block, inside = None, False
    for line in open(DOC).read().split("\n"):
        if line.startswith("```"):
            if inside:
                if block and marker in block[0]:
                    return "\n".join(block) + "\n"
                inside, block = False, None
            else:
                inside, block = True, []
        elif inside:
            block.append(line)
    raise AssertionError(f"INTEGRATION.md has no code block starting with {marker!r}")


def _link_args():
    return ["-L", CSRC, "-lbrotlig_hip", "-lbrotlig_cpu", "-Wl,-rpath," + CSRC, "-pthread"]


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    _build.build_hip(); _build.build_cpu()
    return tmp_path_factory.mktemp("shims")


def test_decode_gpu_shim_of_the_integration_guide_compiles_and_links(built):
    shim = _doc_block("BrotligGPUDecoder_hip.cpp")
    # the reference header the shim includes, reduced to the declaration it implements (sample/BrotligGPUDecoder.h:24);
    # BROTLIG_ERROR comes from this repo's header, whose enumerators equal inc/common/BrotligCommon.h:50-68
    (built / "BrotligGPUDecoder.h").write_text(
        '#pragma once\n#include <cstdint>\n#include "brotlig_amd.h"\n'
        "BROTLIG_ERROR DecodeGPU(bool useWarpDevice, uint32_t input_size, const uint8_t* input, uint32_t* output_size, uint8_t* output, double& time);\n")
    (built / "shim.cpp").write_text(shim)
    (built / "main.cpp").write_text(
        '#include "BrotligGPUDecoder.h"\n#include <vector>\n'
        "int main(int argc, char**) { std::vector<uint8_t> in(16), out(16); uint32_t n = 16; double t = 0;\n"
        "  return argc > 5 ? (int)DecodeGPU(false, 16, in.data(), &n, out.data(), t) : 0; }\n")
    exe = built / "shim_test"
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", str(built), "-I", INC, str(built / "shim.cpp"), str(built / "main.cpp"),
                           "-o", str(exe)] + _link_args())
    subprocess.check_call([str(exe)])                               # loads the libraries, calls nothing on a device


def test_public_headers_are_plain_c(built):
    (built / "c_user.c").write_text(
        '#include "brotlig_amd.h"\n#include "brotlig_amd_cpu.h"\n'
        "static int progress(int type, const char* msg, void* user) { (void)type; (void)msg; (void)user; return 0; }\n"
        "int main(int argc, char** argv) { (void)argv; BrotligDeviceBatch b; BrotligStreamDesc d; (void)b; (void)d;\n"
        "  uint32_t first[3]; uint64_t sizes[2] = {10, 20};\n"
        "  if (BrotligShardPlan(sizes, 2, 2, first) != BROTLIG_OK || first[1] != 1u) return 2;\n"
        "  if (BrotligAbiVersion() != BROTLIG_AMD_ABI_VERSION) return 3;\n"
        "  if (argc > 5) { uint32_t n = 0; return (int)BrotligDecodeCPUWithFeedback(0, 0, &n, 0, 0, progress, 0); }\n"
        "  return 0; }\n")
    exe = built / "c_user"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", INC, str(built / "c_user.c"), "-o", str(exe)] + _link_args())
    subprocess.check_call([str(exe)])


MULTI_C = os.path.join(ROOT, "tools", "multi_device_example.c")


def test_multi_device_example_compiles_against_the_c_abi(built):
    """tools/multi_device_example.c: the C host INTEGRATION.md section 5 describes (BrotligShardPlan + one BrotligDeviceBatch
    per device + BrotligDecodeBatchMultiDevice), built with hipcc as a plain C++/HIP host program."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = os.path.join(ROOT, "tools", "multi_device_example")
    subprocess.check_call([hipcc, "-x", "hip", "--offload-arch=gfx950", "-O1", "-I", INC, MULTI_C, "-o", exe] + _link_args()[:-1] +
                          ["-L", CSRC, "-lbrotlig_enc"])
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_multi_device_example_runs(built):
    exe = os.path.join(ROOT, "tools", "multi_device_example")
    if not os.path.exists(exe):
        test_multi_device_example_compiles_against_the_c_abi(built)
    r = subprocess.run([exe, "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bit-exact" in r.stdout and "named as the damaged one" in r.stdout


def test_decode_cpu_feedback_trampoline_of_the_integration_guide_runs(built):
    """the reference's C++ DecodeCPU signature (std::string callback) on top of the C twin, through a
    trampoline.  Compiled from the guide's text and RUN here (CPU only): one progress call per page, abort honoured."""
    tramp = _doc_block("reference side: DecodeCPU with a working feedbackProc")
    (built / "cpu_shim.cpp").write_text(
        '#include <string>\n#include <cstdio>\n#include <cstring>\n#include <vector>\n#include <atomic>\n'
        '#include "brotlig_amd_cpu.h"\n#include "brotlig_encoder.h"\n'
        # what the reference's headers provide (inc/common/BrotligCommon.h:70-92), minus what brotlig_amd_cpu.h already declares
        "typedef bool (*BROTLIG_Feedback_Proc)(BROTLIG_MESSAGE_TYPE type, std::string message);\n"
        "namespace BrotliG { BROTLIG_ERROR DecodeCPU(uint32_t, const uint8_t*, uint32_t*, uint8_t*, BROTLIG_Feedback_Proc); }\n"
        + tramp +
        "static std::atomic<int> calls{0}; static int stop_after = 0;\n"
        "static bool progress(BROTLIG_MESSAGE_TYPE t, std::string m) { (void)m; return t == BROTLIG_PROGRESS && ++calls >= stop_after && stop_after; }\n"
        "int main() {\n"
        "  std::vector<uint8_t> src(5 * 65536 + 123); for (size_t i = 0; i < src.size(); ++i) src[i] = (uint8_t)((i * 2654435761u) >> 13) & 0x3F;\n"
        "  uint32_t cap = BrotligEncMaxCompressedSize((uint32_t)src.size(), 65536); std::vector<uint8_t> enc(cap);\n"
        "  BrotligEncodeOptions opt; memset(&opt, 0, sizeof opt);\n"
        "  if (BrotligEncode((uint32_t)src.size(), src.data(), &cap, enc.data(), &opt) != 0) return 2;\n"
        "  std::vector<uint8_t> out(src.size()); uint32_t n = (uint32_t)out.size();\n"
        "  if (BrotliG::DecodeCPU(cap, enc.data(), &n, out.data(), progress) != BROTLIG_OK || n != src.size() || memcmp(out.data(), src.data(), n)) return 3;\n"
        "  if (calls != 6) return 4;                      // six pages, one message each\n"
        "  calls = 0; stop_after = 2; n = (uint32_t)out.size();\n"
        "  if (BrotliG::DecodeCPU(cap, enc.data(), &n, out.data(), progress) != BROTLIG_ABORTED) return 5;\n"
        "  n = (uint32_t)out.size();\n"
        "  return BrotliG::DecodeCPU(cap, enc.data(), &n, out.data(), nullptr) == BROTLIG_OK ? 0 : 6;\n}\n")
    exe = built / "cpu_shim"
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", INC, "-I", CSRC, str(built / "cpu_shim.cpp"), "-o", str(exe),
                           "-L", CSRC, "-lbrotlig_cpu", "-lbrotlig_enc", "-Wl,-rpath," + CSRC, "-pthread"])
    subprocess.check_call([str(exe)])
