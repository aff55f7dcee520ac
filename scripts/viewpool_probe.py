#!/usr/bin/env python3
"""Timing probe of the fused view-pooling kernel at the released configuration's shapes
(configs/apple.yaml:166-196: ResNet34 stages 1-4 projected to 16 channels at 64^2 / 32^2 / 16^2 / 8^2 for a 256^2 input,
+ mask (1 ch) + image (3 ch) at 256^2 -> aggregated 2 * 68 = 136 features -> pooled_feature_mapper -> tanh), 16 source
views (the kernel's per-call maximum), onto the 64^3 x 32 grid.  usage: python scripts/viewpool_probe.py [n_src=16] [R=64]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import holo_diffusion_amd as hda  # noqa: E402

n_src = int(sys.argv[1]) if len(sys.argv) > 1 else 16
R = int(sys.argv[2]) if len(sys.argv) > 2 else 64
F = 32
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
feats = {f"res{i}": torch.randn(n_src, 16, s, s, generator=g).tanh() for i, s in enumerate((64, 32, 16, 8))}
feats["mask"] = torch.rand(n_src, 1, 256, 256, generator=g)
feats["rgb"] = torch.rand(n_src, 3, 256, 256, generator=g)
A = 2 * sum(v.shape[1] for v in feats.values())
model = hda.HoloDiffusionModel(resol=R, feature_size=F, view_pooler_enabled=True, net_3d_enabled=False,
                               diffusion_enabled=False, render_image_width=8, render_image_height=8)
model.load_state_dict({"pooled_feature_mapper.weight": 0.1 * torch.randn(F, A, generator=g),
                       "pooled_feature_mapper.bias": torch.zeros(F)}, strict=False)
model.to(dev)
cams = hda.get_simple_360_camera_trajectory(2 * math.pi, n_src, -0.5, 10, (0.0, -1.0, 0.0), 3.2).to(dev)
dfeats = {k: v.to(dev) for k, v in feats.items()}
for _ in range(3):
    out = model.pool_views_to_voxel_features(dfeats, cams)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 20
e0.record()
for _ in range(iters):
    out = model.pool_views_to_voxel_features(dfeats, cams)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
vox = R ** 3
chan = sum(v.shape[1] for v in feats.values())
logical = vox * n_src * chan * 4 * 4  # bilinear taps x 4 bytes
maps = sum(v.numel() for v in feats.values()) * 4
print(f"view pooling: {n_src} views -> {R}^3 x {F}: {ms:.3f} ms per call (incl. the host wrapper); aggregated features {A}; "
      f"logical gather {logical / 1e9:.2f} GB -> {logical / ms / 1e6:.0f} GB/s; feature maps {maps / 1e6:.1f} MB (cache resident); "
      f"output {vox * F * 4 / 1e6:.1f} MB; finite {bool(torch.isfinite(out).all())}")

# ---- backward of the same call (holo_view_pool_backward), and of the learnt aggregator's path (holo_mlp_mean_backward)
gout = torch.randn(1, F, R, R, R, device=dev)
for _ in range(2):
    gr = model.pool_views_backward(dfeats, cams, gout)
torch.cuda.synchronize()
e0.record()
for _ in range(5):
    gr = model.pool_views_backward(dfeats, cams, gout)
e1.record()
torch.cuda.synchronize()
print(f"view pooling backward (AngleWeightedReduction): {e0.elapsed_time(e1) / 5:.3f} ms per call; "
      f"finite {all(bool(torch.isfinite(v).all()) for v in gr['image_features'].values())}")
m2 = hda.HoloDiffusionModel(resol=R, feature_size=F, view_pooler_enabled=True, net_3d_enabled=False, diffusion_enabled=False,
                            render_image_width=8, render_image_height=8,
                            view_pooler_args=dict(feature_aggregator_class_type="MLPMeanFeatureAggregator"))
m2.load_state_dict({"pooled_feature_mapper.weight": 0.1 * torch.randn(F, 128, generator=g),
                    "pooled_feature_mapper.bias": torch.zeros(F)}, strict=False)
m2.to(dev)
n2 = min(n_src, 4)
f2 = {k: v[:n2] for k, v in dfeats.items()}
c2 = cams[list(range(n2))]
for _ in range(2):
    out2 = m2.pool_views_to_voxel_features(f2, c2)
torch.cuda.synchronize()
e0.record()
for _ in range(5):
    out2 = m2.pool_views_to_voxel_features(f2, c2)
e1.record()
torch.cuda.synchronize()
fwd_ms = e0.elapsed_time(e1) / 5
m2.pool_views_backward(f2, c2, gout)
torch.cuda.synchronize()
e0.record()
for _ in range(3):
    gr2 = m2.pool_views_backward(f2, c2, gout)
e1.record()
torch.cuda.synchronize()
print(f"MLPMean aggregator, {n2} views: forward {fwd_ms:.3f} ms, backward {e0.elapsed_time(e1) / 3:.3f} ms per call "
      f"(row buffers + gemm_kernel; workspace {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB peak allocated)")
