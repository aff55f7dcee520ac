#!/bin/bash
# Builds tools/attn_probe (the library's kernels_attn_bf16.o + the driver).
set -e
cd "$(dirname "$0")/.."
H=/opt/rocm/bin/hipcc
make -C holo_diffusion_amd/csrc kernels_attn_bf16.o kernels_attn_bf16_lazy.o > /dev/null
$H -O2 --offload-arch=gfx950 -Wno-unused-result -c tools/attn_probe.cpp -o /tmp/attn_probe.o
$H --offload-arch=gfx950 /tmp/attn_probe.o holo_diffusion_amd/csrc/kernels_attn_bf16.o holo_diffusion_amd/csrc/kernels_attn_bf16_lazy.o -o tools/attn_probe
