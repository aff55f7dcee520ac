"""TEST-ONLY launcher of `holo_diffusion_amd.generate.main` with a stand-in for `load_experiment`: the model is a few
torch-CPU lines (a seeded "sample" and frames that are a deterministic function of sample and camera), so the multi-rank
product entry - rendezvous, device selection, sample sharding, per-sample seeds, frame all_gather, rank-0 output stage -
runs end to end over gloo in the GPU-less container within seconds.  (The kernels themselves are covered by the
`-m gpu` tests; nothing in the package can reach this file.)"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)

from holo_diffusion_amd.checkpoint import LoadReport, model_args_from_expconfig, read_expconfig  # noqa: E402
from holo_diffusion_amd.generate import main  # noqa: E402


class StubModel(torch.nn.Module):
    net_3d_enabled = True
    diffusion_enabled = True

    def __init__(self, kw):
        super().__init__()
        self.render_image_width, self.render_image_height = kw["render_image_width"], kw["render_image_height"]
        self.shape = (1, kw["feature_size"]) + (kw["resol"],) * 3

    def sample_random_voxel_features(self, **kw):
        return torch.randn(self.shape)  # drawn from the per-sample seed generate_samples sets

    def render_views(self, vf, cams):
        n, H, W = len(cams), self.render_image_height, self.render_image_width
        base = vf.mean() + cams.T.sum(dim=1).cpu()  # depends on the sample AND the camera
        img = base[:, None, None, None] + torch.arange(3 * H * W, dtype=torch.float32).reshape(1, 3, H, W) / (3 * H * W)
        return {"images_render": img, "depths_render": img[:, :1] * 2.0, "masks_render": (img[:, :1] > 0).float()}


def load_fn(exp_dir, render_size=None, device=None):
    cfg, fn = read_expconfig(exp_dir)
    kw, ignored = model_args_from_expconfig(cfg, render_size)
    return StubModel(kw), LoadReport(config_file=fn, ignored_config_fields=ignored)


if __name__ == "__main__":
    raise SystemExit(main(sys.argv[1:], load_fn=load_fn))
