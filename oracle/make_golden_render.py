#!/usr/bin/env python3
"""Golden vectors for the parts of the RENDER half of the reference that can be executed without PyTorch3D.

Runs ONLY in the development container (``/root/reference`` present).  Every render-side file of the reference imports
PyTorch3D at module top, so none of them can be imported; but three pieces are plain torch code:

  * ``MLPWithInputSkips``            holo_diffusion/custom_modules.py:44-160   (incl. the :108-112 construction quirk:
                                     the hidden activation lands on the LAST layer only)
  * ``RenderMLP``                    holo_diffusion/holo_voxel_grid_implicit_function.py:48-129
  * ``_make_shaded_from_normals``    holo_diffusion/utils/render_utils/flyaround.py:400-420

This script takes their source text out of the reference files with ``ast`` (nothing is copied into the repository),
executes exactly those class / function bodies, and records inputs and outputs into ``tests/golden/ref_render_mlp.npz``
and ``tests/golden/ref_shaded_from_normals.npz``.  What stands in for PyTorch3D while doing so (and therefore stays
UNPINNED) is only scaffolding:

  * ``Configurable``       -> ``dataclasses.dataclass(eq=False)`` + ``torch.nn.Module.__init__`` before the fields are set
                              (what PyTorch3D's ``expand_args_fields`` generates for a Module)
  * ``DecoderActivation``  -> an Enum with the four members the class body looks up
  * ``_xavier_init``       -> ``torch.nn.init.xavier_uniform_`` (irrelevant: every tensor is overwritten by synthetic weights)
  * ``HarmonicEmbedding``  -> the oracle's restatement (``oracle.render_oracle.harmonic_embedding``: sin | cos | input,
                              frequencies 2^k) - the direction-embedding ORDER therefore remains unpinned

Usage:  python oracle/make_golden_render.py
"""
from __future__ import annotations

import ast
import dataclasses
import enum
import os
import sys
from typing import Dict, Optional, Tuple, Union  # noqa: F401  (names used by the executed reference source)

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np
import torch
import torch.nn.functional as F  # noqa: F401

from holo_diffusion_amd.weights import synth_state_dict  # noqa: E402
from oracle import render_oracle as ro  # noqa: E402
from oracle.common import np_noise  # noqa: E402

REF = "/root/reference/holo_diffusion"
GOLD = os.path.join(REPO, "tests", "golden")


def source_of(path: str, names) -> str:
    """Source text of the top-level classes / functions ``names`` of a reference file, in file order."""
    text = open(path).read()
    tree = ast.parse(text)
    out = []
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name in names:
            out.append(ast.get_source_segment(text, node))
    assert len(out) == len(names), (path, names, len(out))
    return "\n\n".join(out)


class DecoderActivation(enum.Enum):  # pytorch3d.implicitron.models.implicit_function.decoding_functions
    RELU = "relu"
    SOFTPLUS = "softplus"
    SIGMOID = "sigmoid"
    IDENTITY = "identity"


class HarmonicEmbedding(torch.nn.Module):  # pytorch3d.renderer.HarmonicEmbedding(n, logspace=True, append_input=True)
    def __init__(self, n_harmonic_functions: int = 6):
        super().__init__()
        self.n = n_harmonic_functions

    def forward(self, x):
        return ro.harmonic_embedding(x, self.n) if self.n > 0 else x

    def get_output_dim(self, input_dims: int = 3) -> int:
        return input_dims * (2 * self.n + 1)


class Configurable:  # marker only; see configurable() below
    pass


def configurable(cls):
    """What PyTorch3D's expand_args_fields does for a Configurable Module: dataclass fields from the annotations, the
    Module initialised before the fields are assigned, __post_init__ afterwards."""
    cls = dataclasses.dataclass(eq=False)(cls)
    dc_init = cls.__init__

    def init(self, *a, **k):
        torch.nn.Module.__init__(self)
        dc_init(self, *a, **k)

    cls.__init__ = init
    return cls


def build_namespace() -> dict:
    ns = dict(torch=torch, F=F, Enum=enum.Enum, Optional=Optional, Tuple=Tuple, Union=Union, Dict=Dict,
              Configurable=Configurable, DecoderActivation=DecoderActivation, HarmonicEmbedding=HarmonicEmbedding,
              _xavier_init=lambda lin: torch.nn.init.xavier_uniform_(lin.weight.data), COLOUR_DIMS=3)
    exec(source_of(os.path.join(REF, "custom_modules.py"), ["HiddenActivation", "MLPWithInputSkips"]), ns)
    ns["MLPWithInputSkips"] = configurable(ns["MLPWithInputSkips"])
    exec(source_of(os.path.join(REF, "holo_voxel_grid_implicit_function.py"), ["RenderMLP"]), ns)
    ns["RenderMLP"] = configurable(ns["RenderMLP"])
    exec(source_of(os.path.join(REF, "utils", "render_utils", "flyaround.py"), ["_make_shaded_from_normals"]), ns)
    return ns


def main():
    ns = build_namespace()
    out: Dict[str, np.ndarray] = {}
    for C in (16, 32):
        # HoloDiffusionModel forces feature_dim = 0 (holo_diffusion_model.py:156) -> output_vp_independent_feature_dims = 0
        mlp = ns["RenderMLP"](input_dims=C, output_vp_independent_feature_dims=0).eval()
        rcfg = ro.RenderCfg(feature_size=C)
        shapes = ro.render_mlp_param_shapes(rcfg)
        ref_shapes = {k: tuple(v.shape) for k, v in mlp.state_dict().items()}
        assert ref_shapes == {k: tuple(v) for k, v in shapes.items()}, (ref_shapes, shapes)  # the reference's key names
        # the construction quirk, as built by the reference's own loop: LeakyReLU after the LAST density layer only
        acts = [type(layer[1]).__name__ for layer in mlp._density_net.mlp]
        assert acts == ["Identity", "Identity", "Identity", "LeakyReLU"], acts
        sd = synth_state_dict(shapes, 4321 + C)
        sd["_density_net.mlp.3.0.bias"][-1] += 0.05
        mlp.load_state_dict(sd)
        feats = torch.tanh(torch.from_numpy(np_noise(100 + C, (3, 7, 11, C))))
        dirs = torch.nn.functional.normalize(torch.from_numpy(np_noise(200 + C, (3, 7, 3))), dim=-1)
        dirs = dirs[..., None, :].expand(3, 7, 11, 3).contiguous()
        with torch.no_grad():
            dens, col, extra = mlp(feats, dirs)
        assert extra is None
        o_dens, o_col = ro.render_mlp(sd, feats, dirs, rcfg)
        assert torch.allclose(o_dens, dens, rtol=0, atol=2e-6) and torch.allclose(o_col, col, rtol=0, atol=2e-6), \
            ((o_dens - dens).abs().max(), (o_col - col).abs().max())
        ones = torch.nn.functional.normalize(torch.ones_like(dirs), dim=-1)  # the "dummy directions" of the pts_3d entry
        with torch.no_grad():                                                   # (holo_voxel_grid_implicit_function.py:229-239)
            dens1, col1, _ = mlp(feats, ones)
        out[f"C{C}.densities_ones_dir"] = dens1.numpy()
        out[f"C{C}.colours_ones_dir"] = col1.numpy()
        out[f"C{C}.features"] = feats.numpy()
        out[f"C{C}.dirs"] = dirs.numpy()
        out[f"C{C}.densities"] = dens.numpy()
        out[f"C{C}.colours"] = col.numpy()
        out[f"C{C}.seed"] = np.array(4321 + C)
        print(f"RenderMLP C={C}: oracle vs reference class max|d| dens {float((o_dens - dens).abs().max()):.2e} "
              f"colour {float((o_col - col).abs().max()):.2e}")
    np.savez_compressed(os.path.join(GOLD, "ref_render_mlp.npz"), **out)

    n = torch.nn.functional.normalize(torch.from_numpy(np_noise(7, (4, 3, 9, 13))), dim=1) * 0.9
    mask = torch.sigmoid(3.0 * torch.from_numpy(np_noise(8, (1, 1, 9, 13))))
    shaded = ns["_make_shaded_from_normals"](n, mask)
    np.savez_compressed(os.path.join(GOLD, "ref_shaded_from_normals.npz"), normals=n.numpy(), mask=mask.numpy(),
                        shaded=shaded.numpy())
    from holo_diffusion_amd.flyaround_output import make_shaded_from_normals
    assert torch.equal(make_shaded_from_normals(n, mask), shaded)
    print("shaded-from-normals: bit-equal; fixtures written to", GOLD)


if __name__ == "__main__":
    main()
