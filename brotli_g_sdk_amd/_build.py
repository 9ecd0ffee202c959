"""In-tree native builds.  Everything is compiled with explicit commands (no JIT cache) so the
resulting .so files travel with the repo snapshot to the GPU box."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)

ENC_SO = os.path.join(CSRC, "libbrotlig_enc.so")
HIP_SO = os.path.join(CSRC, "libbrotlig_hip.so")
CPU_SO = os.path.join(CSRC, "libbrotlig_cpu.so")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_encoder(force=False):
    src = [os.path.join(CSRC, "brotlig_encoder.cpp"), os.path.join(CSRC, "brotlig_encoder.h")]
    if force or _stale(ENC_SO, src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", ENC_SO, src[0]], cwd=CSRC)
    return ENC_SO


def build_cpu(force=False):
    """DecodeCPU of the reference API (inc/BrotligDecoder.h:33): its own library, never loaded by the GPU path."""
    src = [os.path.join(CSRC, "brotlig_cpu.cpp"), os.path.join(CSRC, "brotlig_format.h"), os.path.join(CSRC, "brotlig_shard_plan.h"),
           os.path.join(ROOT, "include", "brotlig_amd.h"), os.path.join(ROOT, "include", "brotlig_amd_cpu.h")]
    if force or _stale(CPU_SO, src):
        subprocess.check_call(["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-I", os.path.join(ROOT, "include"),
                               "-I", CSRC, "-o", CPU_SO, src[0]], cwd=CSRC)
    return CPU_SO


def hip_sources():
    """The two translation units first, then every header of csrc/ they may include (the kernels are a set of per-stage headers under
    brotlig_kernels.h)."""
    import glob
    units = [os.path.join(CSRC, n) for n in ("brotlig_hip.hip", "brotlig_streamer.hip")]
    headers = sorted(h for h in glob.glob(os.path.join(CSRC, "brotlig_*.h")) if not h.endswith("brotlig_encoder.h"))
    return units + headers + [os.path.join(ROOT, "include", "brotlig_amd.h")]


def kernel_headers():
    """The headers the kernels are made of (hashed by bench.py, watched by the simulator build of tests/test_sim_decode.py)."""
    import glob
    skip = ("brotlig_encoder.h", "brotlig_internal.h", "brotlig_shard_plan.h")
    return sorted(h for h in glob.glob(os.path.join(CSRC, "brotlig_*.h")) if not h.endswith(skip))


# How the device code is compiled, in ONE place (profiles/tools/isa_budget.py, tests/test_bench_gating.py and the shell tools under
# profiles/tools build variants of the kernels with the same flags).  -fno-unroll-loops since round 6: the loop unroller's own choices cost
# the page kernel forty spilled values and a fifth of its code (13 527 -> 10 780 instructions); without them every data class decodes
# 1.0 .. 2.1 % faster, and the round loop has room for the runs-first team levels (profiles/experiments/README.md).  Loops that index
# small register arrays carry `#pragma unroll` (the per-page delta decode), which the flag leaves alone.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-unroll-loops"]


def build_hip(force=False):
    override = os.environ.get("BROTLIG_HIP_SO")     # diagnostics: load an alternative build of the library
    if override:
        return override
    src = hip_sources()
    if force or _stale(HIP_SO, src):
        cmd = [_hipcc()] + HIP_FLAGS + ["-fPIC", "-shared",
               "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", HIP_SO, src[0], src[1]]
        subprocess.check_call(cmd, cwd=CSRC)
    return HIP_SO


CLI_BIN = os.path.join(ROOT, "tools", "brotlig")


def build_cli(force=False):
    """The portable command-line tool (tools/brotlig_cli.cpp), linked against the in-tree libraries."""
    src = os.path.join(ROOT, "tools", "brotlig_cli.cpp")
    build_cpu()
    if force or _stale(CLI_BIN, [src, ENC_SO, HIP_SO, CPU_SO]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", CLI_BIN, src,
                               "-L", CSRC, "-lbrotlig_enc", "-lbrotlig_hip", "-lbrotlig_cpu", "-pthread", "-Wl,-rpath,$ORIGIN/../brotli_g_sdk_amd/csrc"])
    return CLI_BIN


def build_all(force=False):
    build_encoder(force)
    build_hip(force)
    build_cpu(force)
    build_cli(force)
