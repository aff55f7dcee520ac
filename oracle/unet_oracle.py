"""CPU restatement (torch CPU ops, fp32) of the reference 3D UNet denoiser.

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  Functional form: the network
is described by a ``UNetCfg`` and a flat ``state_dict`` that uses the reference's
parameter names, so the same dict loads into the reference ``UNetModel`` (in the
development container) and drives this restatement and the HIP path.

Follows (all paths relative to /root/reference):
  * ``holo_diffusion/utils/diffusion_utils.py:41-86``  SimpleUnet3D ctor mapping
  * ``holo_diffusion/guided_diffusion/unet.py:597-798`` UNetModel block construction
  * ``holo_diffusion/guided_diffusion/unet.py:800-837`` UNetModel.forward
  * ``holo_diffusion/guided_diffusion/unet.py:236-256`` ResBlock._forward (scale-shift norm)
  * ``holo_diffusion/guided_diffusion/unet.py:396-406,436-455`` AttentionBlock / QKVAttentionLegacy
  * ``holo_diffusion/guided_diffusion/unet.py:92-105,129-138`` Upsample / Downsample
  * ``holo_diffusion/guided_diffusion/nn.py:23-25,99-127`` GroupNorm32, timestep_embedding
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class UNetCfg:
    """Mirror of SimpleUnet3D's config fields (diffusion_utils.py:43-53)."""
    image_size: int = 64
    in_channels: int = 128
    out_channels: int = 128
    model_channels: int = 128
    num_res_blocks: int = 2
    channel_mult: Tuple[int, ...] = (1, 2, 4, 8)
    attention_resolutions: Tuple[int, ...] = (8, 16)
    num_heads: int = 2
    dropout: float = 0.0
    homogeneous_resample: bool = True


# ----------------------------------------------------------------------------
# structure enumeration (unet.py:645-798)
# ----------------------------------------------------------------------------
@dataclass
class _Block:
    kind: str                 # 'conv' | 'res' | 'attn' | 'down' | 'up'
    prefix: str
    cin: int = 0
    cout: int = 0


def unet_structure(cfg: UNetCfg):
    """Return (input_blocks, middle, output_blocks, out_ch): lists of lists of _Block."""
    mc = cfg.model_channels
    ch = int(cfg.channel_mult[0] * mc)
    inputs: List[List[_Block]] = [[_Block("conv", "input_blocks.0.0", cfg.in_channels, ch)]]
    chans = [ch]
    ds = 1
    idx = 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            layers = [_Block("res", f"input_blocks.{idx}.0", ch, int(mult * mc))]
            ch = int(mult * mc)
            if ds in cfg.attention_resolutions:
                layers.append(_Block("attn", f"input_blocks.{idx}.1", ch, ch))
            inputs.append(layers)
            chans.append(ch)
            idx += 1
        if level != len(cfg.channel_mult) - 1:
            inputs.append([_Block("down", f"input_blocks.{idx}.0", ch, ch)])
            chans.append(ch)
            ds *= 2
            idx += 1
    middle = [
        _Block("res", "middle_block.0", ch, ch),
        _Block("attn", "middle_block.1", ch, ch),
        _Block("res", "middle_block.2", ch, ch),
    ]
    outputs: List[List[_Block]] = []
    oidx = 0
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            layers = [_Block("res", f"output_blocks.{oidx}.0", ch + ich, int(mc * mult))]
            ch = int(mc * mult)
            if ds in cfg.attention_resolutions:
                layers.append(_Block("attn", f"output_blocks.{oidx}.{len(layers)}", ch, ch))
            if level and i == cfg.num_res_blocks:
                layers.append(_Block("up", f"output_blocks.{oidx}.{len(layers)}", ch, ch))
                ds //= 2
            outputs.append(layers)
            oidx += 1
    return inputs, middle, outputs, ch


def unet_param_shapes(cfg: UNetCfg) -> Dict[str, Tuple[int, ...]]:
    """{reference state_dict name: shape} for UNetModel built as SimpleUnet3D does."""
    mc = cfg.model_channels
    ted = 4 * mc
    shapes: Dict[str, Tuple[int, ...]] = {
        "time_embed.0.weight": (ted, mc), "time_embed.0.bias": (ted,),
        "time_embed.2.weight": (ted, ted), "time_embed.2.bias": (ted,),
    }

    def add(b: _Block):
        p = b.prefix
        if b.kind == "conv":
            shapes[p + ".weight"] = (b.cout, b.cin, 3, 3, 3)
            shapes[p + ".bias"] = (b.cout,)
        elif b.kind == "res":
            shapes[p + ".in_layers.0.weight"] = (b.cin,)
            shapes[p + ".in_layers.0.bias"] = (b.cin,)
            shapes[p + ".in_layers.2.weight"] = (b.cout, b.cin, 3, 3, 3)
            shapes[p + ".in_layers.2.bias"] = (b.cout,)
            shapes[p + ".emb_layers.1.weight"] = (2 * b.cout, ted)
            shapes[p + ".emb_layers.1.bias"] = (2 * b.cout,)
            shapes[p + ".out_layers.0.weight"] = (b.cout,)
            shapes[p + ".out_layers.0.bias"] = (b.cout,)
            shapes[p + ".out_layers.3.weight"] = (b.cout, b.cout, 3, 3, 3)
            shapes[p + ".out_layers.3.bias"] = (b.cout,)
            if b.cin != b.cout:
                shapes[p + ".skip_connection.weight"] = (b.cout, b.cin, 1, 1, 1)
                shapes[p + ".skip_connection.bias"] = (b.cout,)
        elif b.kind == "attn":
            shapes[p + ".norm.weight"] = (b.cin,)
            shapes[p + ".norm.bias"] = (b.cin,)
            shapes[p + ".qkv.weight"] = (3 * b.cin, b.cin, 1)
            shapes[p + ".qkv.bias"] = (3 * b.cin,)
            shapes[p + ".proj_out.weight"] = (b.cin, b.cin, 1)
            shapes[p + ".proj_out.bias"] = (b.cin,)
        elif b.kind == "down":
            shapes[p + ".op.weight"] = (b.cout, b.cin, 3, 3, 3)
            shapes[p + ".op.bias"] = (b.cout,)
        elif b.kind == "up":
            shapes[p + ".conv.weight"] = (b.cout, b.cin, 3, 3, 3)
            shapes[p + ".conv.bias"] = (b.cout,)

    inputs, middle, outputs, ch = unet_structure(cfg)
    for layers in inputs:
        for b in layers:
            add(b)
    for b in middle:
        add(b)
    for layers in outputs:
        for b in layers:
            add(b)
    shapes["out.0.weight"] = (ch,)
    shapes["out.0.bias"] = (ch,)
    shapes["out.2.weight"] = (cfg.out_channels, ch, 3, 3, 3)
    shapes["out.2.bias"] = (cfg.out_channels,)
    return shapes


# ----------------------------------------------------------------------------
# ops
# ----------------------------------------------------------------------------
def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """nn.py:109-127."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def group_norm32(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """nn.py:23-25,99-106: GroupNorm(32, C), fp32, eps 1e-5."""
    return F.group_norm(x.float(), 32, w, b, eps=1e-5).type(x.dtype)


def time_embed(sd, cfg: UNetCfg, t: torch.Tensor) -> torch.Tensor:
    """unet.py:645-650, 813."""
    e = timestep_embedding(t, cfg.model_channels)
    e = F.linear(e, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    e = F.silu(e)
    return F.linear(e, sd["time_embed.2.weight"], sd["time_embed.2.bias"])


def res_block(sd, p: str, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """unet.py:236-256 with use_scale_shift_norm=True, no up/down, dropout 0."""
    h = group_norm32(x, sd[p + ".in_layers.0.weight"], sd[p + ".in_layers.0.bias"])
    h = F.silu(h)
    h = F.conv3d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])
    e = e[..., None, None, None]
    scale, shift = torch.chunk(e, 2, dim=1)
    h = group_norm32(h, sd[p + ".out_layers.0.weight"], sd[p + ".out_layers.0.bias"]) * (1 + scale) + shift
    h = F.silu(h)
    h = F.conv3d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if (p + ".skip_connection.weight") in sd:
        x = F.conv3d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


def attention_block(sd, p: str, x: torch.Tensor, num_heads: int) -> torch.Tensor:
    """unet.py:396-406 + QKVAttentionLegacy :436-455 (heads split before q/k/v)."""
    b, c, *spatial = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(group_norm32(xf, sd[p + ".norm.weight"], sd[p + ".norm.bias"]),
                   sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    bs, width, length = qkv.shape
    ch = width // (3 * num_heads)
    q, k, v = qkv.reshape(bs * num_heads, ch * 3, length).split(ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    weight = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    weight = torch.softmax(weight.float(), dim=-1).type(weight.dtype)
    a = torch.einsum("bts,bcs->bct", weight, v).reshape(bs, -1, length)
    h = F.conv1d(a, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (xf + h).reshape(b, c, *spatial)


def downsample(sd, p: str, x: torch.Tensor, homogeneous: bool) -> torch.Tensor:
    """unet.py:123-138 (conv_resample=True)."""
    stride = (2, 2, 2) if homogeneous else (1, 2, 2)
    return F.conv3d(x, sd[p + ".op.weight"], sd[p + ".op.bias"], stride=stride, padding=1)


def upsample(sd, p: str, x: torch.Tensor, homogeneous: bool) -> torch.Tensor:
    """unet.py:92-106: nearest x2 then conv3."""
    if homogeneous:
        size = (x.shape[2] * 2, x.shape[3] * 2, x.shape[4] * 2)
    else:
        size = (x.shape[2], x.shape[3] * 2, x.shape[4] * 2)
    x = F.interpolate(x, size, mode="nearest")
    return F.conv3d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)


def _run_layers(sd, cfg, layers, h, emb, trace, tag):
    for b in layers:
        if b.kind == "conv":
            h = F.conv3d(h, sd[b.prefix + ".weight"], sd[b.prefix + ".bias"], padding=1)
        elif b.kind == "res":
            h = res_block(sd, b.prefix, h, emb)
        elif b.kind == "attn":
            h = attention_block(sd, b.prefix, h, cfg.num_heads)
        elif b.kind == "down":
            h = downsample(sd, b.prefix, h, cfg.homogeneous_resample)
        elif b.kind == "up":
            h = upsample(sd, b.prefix, h, cfg.homogeneous_resample)
    if trace is not None:
        trace[tag] = h
    return h


@torch.no_grad()
def unet_forward(sd: Dict[str, torch.Tensor], cfg: UNetCfg, x: torch.Tensor, timesteps: torch.Tensor,
                 trace: Optional[dict] = None) -> torch.Tensor:
    """UNetModel.forward (unet.py:800-837).  ``trace`` (optional dict) receives every block output."""
    inputs, middle, outputs, _ = unet_structure(cfg)
    emb = time_embed(sd, cfg, timesteps)
    if trace is not None:
        trace["emb"] = emb
    hs = []
    h = x.float()
    for i, layers in enumerate(inputs):
        h = _run_layers(sd, cfg, layers, h, emb, trace, f"input_blocks.{i}")
        hs.append(h)
    h = _run_layers(sd, cfg, middle, h, emb, trace, "middle_block")
    for i, layers in enumerate(outputs):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_layers(sd, cfg, layers, h, emb, trace, f"output_blocks.{i}")
    h = group_norm32(h, sd["out.0.weight"], sd["out.0.bias"])
    h = F.silu(h)
    return F.conv3d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)
