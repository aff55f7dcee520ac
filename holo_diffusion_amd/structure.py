"""Parameter layout of the denoiser, by reference state_dict name.

Host-side restatement of the block construction in
/root/reference/holo_diffusion/guided_diffusion/unet.py:645-798 (as SimpleUnet3D configures it,
utils/diffusion_utils.py:56-75: dims=3, use_scale_shift_norm, conv_resample, no resblock_updown,
num_head_channels=-1).  The C++ planner (csrc/unet_exec.cpp) builds the same list; SimpleUnet3D
cross-checks the two when it creates its native handle.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple


def unet_blocks(model_channels: int, in_channels: int, num_res_blocks: int, channel_mult: Sequence[int],
                attention_resolutions: Sequence[int]):
    """Returns (input_blocks, middle_block, output_blocks, final_ch); each block is a list of
    (kind, prefix, cin, cout) with kind in {conv, res, attn, down, up}."""
    mc = model_channels
    ch = int(channel_mult[0] * mc)
    inputs: List[List[Tuple[str, str, int, int]]] = [[("conv", "input_blocks.0.0", in_channels, ch)]]
    chans = [ch]
    ds, idx = 1, 1
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            layers = [("res", f"input_blocks.{idx}.0", ch, int(mult * mc))]
            ch = int(mult * mc)
            if ds in attention_resolutions:
                layers.append(("attn", f"input_blocks.{idx}.1", ch, ch))
            inputs.append(layers)
            chans.append(ch)
            idx += 1
        if level != len(channel_mult) - 1:
            inputs.append([("down", f"input_blocks.{idx}.0", ch, ch)])
            chans.append(ch)
            ds *= 2
            idx += 1
    middle = [("res", "middle_block.0", ch, ch), ("attn", "middle_block.1", ch, ch), ("res", "middle_block.2", ch, ch)]
    outputs = []
    oidx = 0
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for i in range(num_res_blocks + 1):
            ich = chans.pop()
            layers = [("res", f"output_blocks.{oidx}.0", ch + ich, int(mc * mult))]
            ch = int(mc * mult)
            if ds in attention_resolutions:
                layers.append(("attn", f"output_blocks.{oidx}.{len(layers)}", ch, ch))
            if level and i == num_res_blocks:
                layers.append(("up", f"output_blocks.{oidx}.{len(layers)}", ch, ch))
                ds //= 2
            outputs.append(layers)
            oidx += 1
    return inputs, middle, outputs, ch


def unet_param_shapes(image_size: int, in_channels: int, out_channels: int, model_channels: int,
                      num_res_blocks: int, channel_mult: Sequence[int],
                      attention_resolutions: Sequence[int]) -> Dict[str, Tuple[int, ...]]:
    del image_size
    ted = 4 * model_channels
    s: Dict[str, Tuple[int, ...]] = {
        "time_embed.0.weight": (ted, model_channels), "time_embed.0.bias": (ted,),
        "time_embed.2.weight": (ted, ted), "time_embed.2.bias": (ted,),
    }
    inputs, middle, outputs, ch = unet_blocks(model_channels, in_channels, num_res_blocks, channel_mult,
                                              attention_resolutions)

    def add(kind, p, ci, co):
        if kind == "conv":
            s[p + ".weight"], s[p + ".bias"] = (co, ci, 3, 3, 3), (co,)
        elif kind == "res":
            s[p + ".in_layers.0.weight"], s[p + ".in_layers.0.bias"] = (ci,), (ci,)
            s[p + ".in_layers.2.weight"], s[p + ".in_layers.2.bias"] = (co, ci, 3, 3, 3), (co,)
            s[p + ".emb_layers.1.weight"], s[p + ".emb_layers.1.bias"] = (2 * co, ted), (2 * co,)
            s[p + ".out_layers.0.weight"], s[p + ".out_layers.0.bias"] = (co,), (co,)
            s[p + ".out_layers.3.weight"], s[p + ".out_layers.3.bias"] = (co, co, 3, 3, 3), (co,)
            if ci != co:
                s[p + ".skip_connection.weight"], s[p + ".skip_connection.bias"] = (co, ci, 1, 1, 1), (co,)
        elif kind == "attn":
            s[p + ".norm.weight"], s[p + ".norm.bias"] = (ci,), (ci,)
            s[p + ".qkv.weight"], s[p + ".qkv.bias"] = (3 * ci, ci, 1), (3 * ci,)
            s[p + ".proj_out.weight"], s[p + ".proj_out.bias"] = (ci, ci, 1), (ci,)
        elif kind == "down":
            s[p + ".op.weight"], s[p + ".op.bias"] = (co, ci, 3, 3, 3), (co,)
        elif kind == "up":
            s[p + ".conv.weight"], s[p + ".conv.bias"] = (co, ci, 3, 3, 3), (co,)

    for layers in inputs:
        for b in layers:
            add(*b)
    for b in middle:
        add(*b)
    for layers in outputs:
        for b in layers:
            add(*b)
    s["out.0.weight"], s["out.0.bias"] = (ch,), (ch,)
    s["out.2.weight"], s["out.2.bias"] = (out_channels, ch, 3, 3, 3), (out_channels,)
    return s
