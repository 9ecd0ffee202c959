import os
import sys

import pytest

# the library's two diagnostics switches (which decode kernel, which grid) only work in a process that asks for them in its environment
# BEFORE the library's first launch: the parity matrix below pins each form of the page kernel in turn
os.environ.setdefault("BROTLIG_ENABLE_DEBUG_KNOBS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _hip_device_usable():
    if not os.path.exists("/dev/kfd"):
        return False
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped (not failed) on a machine without a usable HIP device, so that a plain
    `pytest tests` is green in the build container; on the GPU box they run and fail loudly if the HIP
    library is missing (there is no CPU decode path to fall back to)."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu") is not None]
    if not gpu_items or _hip_device_usable():
        return
    if os.environ.get("BROTLIG_REQUIRE_GPU") == "1":
        # on the GPU box a missing device must not pass as a green run of skips
        raise pytest.UsageError("BROTLIG_REQUIRE_GPU=1 but no usable HIP device: the gpu-marked tests cannot run")
    config._brotlig_gpu_skipped = len(gpu_items)
    skip = pytest.mark.skip(reason="no usable HIP device (gpu tests run on the MI355X box)")
    for it in gpu_items:
        it.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    n = getattr(config, "_brotlig_gpu_skipped", 0)
    if n:
        terminalreporter.write_sep("=", f"{n} gpu-marked tests SKIPPED: no HIP device here -- the HIP kernels were exercised by the CPU "
                                        "simulator only (tests/sim); run `pytest -m gpu` on the MI355X box (BROTLIG_REQUIRE_GPU=1 makes "
                                        "a missing device an error)", yellow=True, bold=True)


@pytest.fixture(scope="session")
def oracle():
    from helpers import oracle_lib
    return oracle_lib()


@pytest.fixture(scope="session", autouse=True)
def _decode_mode_from_env():
    """BROTLIG_TEST_DECODE_MODE=1|2 runs the gpu-marked suite with the decode kernel pinned (BrotligDebugSetDecodeMode: 1 = one
    wavefront per one or two pages for every batch, 2 = two wavefronts per page for every batch) instead of the size rule."""
    mode = int(os.environ.get("BROTLIG_TEST_DECODE_MODE", "0"))
    if mode and _hip_device_usable():
        from brotli_g_sdk_amd import api
        api.DebugSetDecodeMode(mode)
    yield


# The forms of the page decode a batch can meet on the device (csrc/brotlig_hip.hip enqueue(), csrc/brotlig_kernels.h decode_kernel_body):
#   rule  -- what the size rule picks; for the small batches of the parity cases that is brotlig_decode_duo_kernel (two wavefronts per page)
#   solo  -- brotlig_decode_kernel, one page per wavefront: decode_pages<.., true>, the one-page window geometry, 64-lane copy teams
#   pair  -- brotlig_decode_kernel forced onto ONE wavefront: decode_pages<.., false>, two pages per wavefront, one per 32-lane half -- the
#            form every large batch (the benchmark) runs.  A batch needs at least two pages for it, so single-stream cases go through
#            BatchDecoder with a companion stream (helpers: decode_all_forms below).
#   pair3 -- the same on THREE wavefronts (round 6): the cases also meet the atomic hand-out of pages between wavefronts,
#            the page schedule's job records taken out of order, and a neighbour wavefront's overflow slots in the workspace.  The batches of
#            this form hold at least six pages, so that every one of the three wavefronts runs the two-page form.
KERNEL_FORMS = (("rule", 0, 0), ("solo", 1, 0), ("pair", 1, 1), ("pair3", 1, 3))


@pytest.fixture(params=KERNEL_FORMS, ids=[f[0] for f in KERNEL_FORMS])
def kernel_form(request):
    """Pins the decode kernel form for one test and puts the size rule back afterwards (the whole oracle-parity matrix on
    every form of the page kernel, under the driver's plain `pytest -m gpu`)."""
    name, mode, grid = request.param
    from brotli_g_sdk_amd import api
    api.DebugSetDecodeMode(mode)
    api.DebugSetDecodeGrid(grid)
    try:
        yield name
    finally:
        api.DebugSetDecodeGrid(0)
        api.DebugSetDecodeMode(int(os.environ.get("BROTLIG_TEST_DECODE_MODE", "0")))
