import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


EMU = os.environ.get("HOLO_TEST_EMU") == "1"


def _enable_emulation():
    """TEST-ONLY (HOLO_TEST_EMU=1): run the `-m gpu` tests of tiny sizes in the GPU-less development container by
    pointing the ctypes binding at tests/emu/libholo_emu.so (the same kernel sources compiled for host threads) and
    letting the plugin classes work on CPU tensors.  Never active in the product or on the GPU box."""
    import ctypes

    import torch

    from holo_diffusion_amd import _lib, runtime
    emu = os.path.join(REPO, "tests", "emu", "libholo_emu.so")
    if not os.path.isfile(emu):
        raise RuntimeError("HOLO_TEST_EMU=1 needs `make -C holo_diffusion_amd/csrc emu`")
    _lib._LIB = _lib.bind(ctypes.CDLL(emu))
    runtime.require_device = lambda *a, **k: None
    runtime.stream_ptr = lambda device=None: ctypes.c_void_p(None)
    runtime.sync_before_destroy = lambda device=None: None
    torch.cuda.current_device = lambda: 0

    class _NoStream:
        cuda_stream = 0

        def synchronize(self):
            pass

    torch.cuda.current_stream = lambda device=None: _NoStream()
    torch.cuda.synchronize = lambda device=None: None


if EMU:
    _enable_emulation()


# Collection order of the test FILES: the hot path of SURVEY.md 8(a)/(b) first - denoiser, sampler, renderer, the sizes
# BASELINE names, the model glue - and the (f) rows (view pooling, training side) last, so that under `pytest -x` a failure on
# the training side can never hide the hot path's evidence (round 5: one training-side assertion stopped the run in front of
# all of test_gpu_unet.py).  No test is dropped; files not listed keep their alphabetical order behind the listed ones.
_FILE_ORDER = ["test_abi_symbols", "test_oracle_golden", "test_render_oracle_kat", "test_host_logic", "test_config_schema",
               "test_gpu_unet", "test_gpu_diffusion", "test_gpu_render", "test_gpu_configs", "test_gpu_model",
               "test_checkpoint_loading", "test_pytorch3d_registration", "test_distributed_cpu", "test_generate_cli",
               "test_viewpool", "test_gpu_training_mode", "test_gpu_backward", "test_gpu_render_backward"]


def pytest_collection_modifyitems(config, items):
    """Orders the files (above), then: gpu-marked tests are SKIPPED (not failed) on a machine without an MI355X or without
    the built library, so a plain `pytest tests` works everywhere; `-m gpu` on the GPU box runs them."""
    rank = {name: i for i, name in enumerate(_FILE_ORDER)}
    stem = lambda item: os.path.splitext(os.path.basename(str(item.fspath)))[0]  # noqa: E731
    items.sort(key=lambda item: (rank.get(stem(item), len(rank)), stem(item)))  # (stable: the order inside a file stays)
    import torch
    lib = os.path.join(REPO, "holo_diffusion_amd", "libholo_mi355x.so")
    if EMU or (torch.cuda.is_available() and os.path.isfile(lib)):
        return
    why = "no HIP device" if not torch.cuda.is_available() else "libholo_mi355x.so is not built"
    skip = pytest.mark.skip(reason=f"gpu test: {why}")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
