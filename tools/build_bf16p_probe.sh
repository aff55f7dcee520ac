#!/bin/bash
# Builds tools/bf16p_probe: the library's conv objects + probe / timeline copies of kernels_conv_bf16p.hip + the driver.
set -e
cd "$(dirname "$0")/.."
H=/opt/rocm/bin/hipcc
F="-O3 -std=c++17 --offload-arch=gfx950 -Iholo_diffusion_amd/csrc -Wno-unused-function"
make -C holo_diffusion_amd/csrc -j8 > /dev/null
O=""
for p in 1 2 3 4 7 8 16 128 256 512; do
  $H $F -DP_PROBE=$p -DP_ENTRY=conv_bf16p_launch_p$p -c holo_diffusion_amd/csrc/kernels_conv_bf16p.hip -o /tmp/bf16p_p$p.o
  O="$O /tmp/bf16p_p$p.o"
done
$H $F -DP_PHASED=1 -DP_ENTRY=conv_bf16p_launch_prio -c holo_diffusion_amd/csrc/kernels_conv_bf16p.hip -o /tmp/bf16p_prio.o
O="$O /tmp/bf16p_prio.o"
$H $F -DP_TIMELINE -DP_ENTRY=conv_bf16p_launch_tl -c holo_diffusion_amd/csrc/kernels_conv_bf16p.hip -o /tmp/bf16p_tl.o
$H -O2 --offload-arch=gfx950 -c tools/bf16p_probe.cpp -o /tmp/bf16p_probe.o
$H --offload-arch=gfx950 /tmp/bf16p_probe.o holo_diffusion_amd/csrc/kernels_conv.o holo_diffusion_amd/csrc/kernels_conv3.o \
   holo_diffusion_amd/csrc/kernels_conv_bf16p.o holo_diffusion_amd/csrc/kernels_misc.o $O /tmp/bf16p_tl.o -o tools/bf16p_probe
