"""CPU: analytic known-answer tests anchoring the (unpinned) renderer half of the oracle
(SURVEY.md §8c item 6): trilinear fetch vs F.grid_sample, emission-absorption closed forms,
sample_pdf on uniform weights, ray geometry."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import render_oracle as ro
from oracle.common import np_noise
from holo_diffusion_amd.weights import synth_state_dict


def test_trilinear_constant_and_impulse():
    cfg = ro.RenderCfg(resol=8, feature_size=4)
    grid = torch.full((1, 4, 8, 8, 8), 0.37)
    half = 0.5 * 7 * (8.0 / 8)
    pts = (torch.rand(500, 3) * 2 - 1) * half * 0.999
    torch.testing.assert_close(ro.trilinear(grid, pts, cfg), torch.full((500, 4), 0.37), rtol=1e-5, atol=1e-6)
    # single-voxel impulse -> trilinear hat weights; x indexes the LAST grid axis
    grid = torch.zeros(1, 1, 8, 8, 8)
    grid[0, 0, 2, 3, 4] = 1.0  # (z=2, y=3, x=4)
    vs = 8.0 / 8
    centre = torch.tensor([[(4 - 3.5) * vs, (3 - 3.5) * vs, (2 - 3.5) * vs]])
    assert abs(ro.trilinear(grid, centre, cfg).item() - 1.0) < 1e-6
    off = centre + torch.tensor([[0.25 * vs, 0.0, 0.0]])
    assert abs(ro.trilinear(grid, off, cfg).item() - 0.75) < 1e-6
    off = centre + torch.tensor([[0.25 * vs, -0.5 * vs, 0.5 * vs]])
    assert abs(ro.trilinear(grid, off, cfg).item() - 0.75 * 0.5 * 0.5) < 1e-6
    # outside the grid by more than one voxel -> 0 (zeros padding)
    assert ro.trilinear(grid, torch.tensor([[10.0, 0.0, 0.0]]), cfg).abs().max() == 0


def test_ea_closed_forms():
    cfg = ro.RenderCfg()
    z = torch.linspace(6.0, 14.0, 64)[None]
    col = torch.rand(1, 64, 3)
    # zero density -> background colour, mask 0
    rgb, depth, mask, w = ro.ea_raymarch(torch.zeros(1, 64, 1), col, z, cfg)
    torch.testing.assert_close(rgb, torch.ones(1, 3))
    assert mask.item() == 0 and depth.item() == 0
    # uniform density sigma on the first k intervals, zero afterwards -> mask = 1 - exp(-sigma * L)
    sigma, k = 0.7, 20
    dens = torch.zeros(1, 64, 1)
    dens[0, :k] = sigma
    rgb, depth, mask, w = ro.ea_raymarch(dens, col, z, cfg)
    L = (z[0, k] - z[0, 0]).item()
    assert abs(mask.item() - (1 - math.exp(-sigma * L))) < 1e-5
    # any positive density at the LAST sample saturates the ray (delta = 1e10)
    dens = torch.zeros(1, 64, 1)
    dens[0, -1] = 1e-3
    _, _, mask, _ = ro.ea_raymarch(dens, col, z, cfg)
    assert mask.item() == 1.0
    # weights sum to the mask
    dens = torch.rand(1, 64, 1)
    dens[0, -1] = 0
    _, _, mask, w = ro.ea_raymarch(dens, col, z, cfg)
    assert abs(w.sum().item() - mask.item()) < 1e-5


def test_sample_pdf_uniform_and_peaked():
    bins = torch.linspace(0.0, 1.0, 63)[None]
    z = ro.sample_pdf(bins, torch.ones(1, 62), 64)
    torch.testing.assert_close(z, torch.linspace(0, 1, 64)[None], rtol=0, atol=2e-6)
    w = torch.zeros(1, 62)
    w[0, 30] = 1.0
    z = ro.sample_pdf(bins, w, 64)
    inside = ((z >= bins[0, 30] - 1e-6) & (z <= bins[0, 31] + 1e-6)).float().mean().item()
    assert inside > 0.9
    assert (z[0, 1:] >= z[0, :-1]).all()


def test_ray_geometry():
    cfg = ro.RenderCfg(image_height=6, image_width=8)
    cams = ro.simple_360_cameras(5)
    cam = {k: v[2:3] for k, v in cams.items()}
    o, d, l = ro.make_rays(cam, cfg)
    R, T = cam["R"][0], cam["T"][0]
    centre = -(T[None] @ R.t())[0]
    torch.testing.assert_close(o, centre[None].expand_as(o), rtol=1e-4, atol=1e-4)
    assert abs(centre.norm().item() - 10.0) < 1e-4
    # camera looks at the origin: the central ray direction is parallel to -centre
    xy = ro.ndc_pixel_grid(6, 8)
    assert xy[0, 0, 0] > 0 and xy[0, 0, 1] > 0 and xy[-1, -1, 0] < 0  # +x left, +y up
    assert abs(l[0, 0].item() - (10.0 - 4.0)) < 1e-4 and abs(l[0, -1].item() - 14.0) < 1e-4
    # world->camera projection of origin + z*dir lands on the pixel's NDC coordinate
    p = o + 7.3 * d
    pc = p @ R + T[None]
    f = cam["focal"][0]
    torch.testing.assert_close(f[0] * pc[:, 0] / pc[:, 2], xy.reshape(-1, 2)[:, 0], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(pc[:, 2], torch.full_like(pc[:, 2], 7.3), rtol=1e-4, atol=1e-4)


def test_collapsed_density_net_matches_uncollapsed():
    """The algebraic fold used by the HIP renderer (render_exec.cpp) equals the layer-by-layer MLP."""
    cfg = ro.RenderCfg(feature_size=32)
    sd = synth_state_dict(ro.render_mlp_param_shapes(cfg), 11)
    f = torch.tanh(torch.from_numpy(np_noise(3, (200, 32))))
    d = F.normalize(torch.from_numpy(np_noise(4, (200, 3))), dim=-1)
    dens, rgb = ro.render_mlp(sd, f, d, cfg)
    W = {k: v.double() for k, v in sd.items()}
    p = "_density_net.mlp."
    A1 = W[p + "1.0.weight"] @ W[p + "0.0.weight"]
    c1 = W[p + "1.0.weight"] @ W[p + "0.0.bias"] + W[p + "1.0.bias"]
    W2 = W[p + "2.0.weight"]
    A2 = W2[:, :256] @ A1 + W2[:, 256:]
    c2 = W2[:, :256] @ c1 + W[p + "2.0.bias"]
    We = W[p + "3.0.weight"] @ A2
    be = W[p + "3.0.weight"] @ c2 + W[p + "3.0.bias"]
    o = F.leaky_relu(f.double() @ We.t() + be, 0.2)
    torch.testing.assert_close(o[:, -1:].float(), dens, rtol=1e-4, atol=1e-5)


def test_normals_of_a_linear_density_field_are_constant():
    """KAT for the oracle's get_normals restatement: a grid whose (single effective) feature is linear in x gives a
    density with constant gradient direction +-x inside the volume."""
    import torch
    from oracle import render_oracle as ro
    from holo_diffusion_amd.weights import synth_state_dict
    R, C = 8, 16
    cfg = ro.RenderCfg(resol=R, feature_size=C, image_height=4, image_width=4)
    sd = synth_state_dict(ro.render_mlp_param_shapes(cfg), 11)
    grid = torch.zeros(1, C, R, R, R)
    grid[0, 0] = torch.linspace(-1, 1, R)[None, None, :].expand(R, R, R)  # varies along x (last axis) only
    pts = (torch.rand(64, 3, generator=torch.Generator().manual_seed(1)) - 0.5) * 6.0  # well inside +-3.5
    n = ro.implicit_normals(grid, sd, pts, cfg)
    assert torch.allclose(n.norm(dim=-1), torch.ones(64), atol=1e-5)
    assert n[:, 1:].abs().max() < 1e-5            # no y / z component
    assert (n[:, 0] - n[0, 0]).abs().max() < 1e-5  # the same sign everywhere
