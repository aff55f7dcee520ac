"""TEST-ONLY launcher of `holo_diffusion_amd.generate.main` on the HOST EMULATION of the kernels (tests/emu): the REAL
model - SimpleUnet3D, the DDPM sampler, the renderer, loaded from an experiment directory - behind the multi-rank product
entry, so that rendezvous, sample sharding, per-sample seeds and the frame all_gather run end to end with the library's own
arithmetic in the GPU-less container (over gloo; on the GPU node the same entry runs over RCCL).  Slow: opt-in
(tests/test_generate_cli.py::test_generate_cli_world2_gloo_on_the_emulated_kernels).  Nothing in the package can reach it."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ["HOLO_TEST_EMU"] = "1"

import tests.conftest  # noqa: E402,F401  (HOLO_TEST_EMU=1: binds the emulation library, lets the plugin work on CPU tensors)
from holo_diffusion_amd import generate as gen  # noqa: E402

# HOLO_TEST_MAX_ITER=<n> (test knob of THIS launcher): every sampling chain is subsampled to n of its steps
# (p_sample_loop's own max_iter, gaussian_diffusion.py:608-621) so that the emulated end-to-end run fits a CPU suite
if os.environ.get("HOLO_TEST_MAX_ITER"):
    import functools

    from holo_diffusion_amd.diffusion import ImplicitronGaussianDiffusion as _D
    _orig = _D.p_sample_loop_progressive

    @functools.wraps(_orig)
    def _capped(self, *a, **k):
        if k.get("max_iter") is None:
            k["max_iter"] = int(os.environ["HOLO_TEST_MAX_ITER"])
        return _orig(self, *a, **k)
    _D.p_sample_loop_progressive = _capped

if __name__ == "__main__":
    # (gen.init_distributed picks gloo + CPU tensors by itself when no HIP device is visible)
    raise SystemExit(gen.main(sys.argv[1:]))
