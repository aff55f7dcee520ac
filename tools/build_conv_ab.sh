#!/bin/bash
# Builds tools/conv_ab: the library's conv objects + probe copies of kernels_conv3.hip (W3_PROBE bits) + the driver.
set -e
cd "$(dirname "$0")/.."
H=/opt/rocm/bin/hipcc
F="-O3 -std=c++17 --offload-arch=gfx950 -Iholo_diffusion_amd/csrc"
make -C holo_diffusion_amd/csrc -j8 > /dev/null
for p in 1 2 4 7 8 16 32 64 128; do
  $H $F -DW3_PROBE=$p -DW3_ENTRY=conv_wino3_launch_p$p -c holo_diffusion_amd/csrc/kernels_conv3.hip -o /tmp/k3_p$p.o
done
$H $F -DW3_TIMELINE -DW3_ENTRY=conv_wino3_launch_tl -c holo_diffusion_amd/csrc/kernels_conv3.hip -o /tmp/k3_tl.o
$H -O2 --offload-arch=gfx950 -c tools/conv_ab.cpp -o /tmp/conv_ab.o
$H --offload-arch=gfx950 /tmp/conv_ab.o holo_diffusion_amd/csrc/kernels_conv.o holo_diffusion_amd/csrc/kernels_conv3.o holo_diffusion_amd/csrc/kernels_conv_bf16p.o \
   holo_diffusion_amd/csrc/kernels_misc.o /tmp/k3_p1.o /tmp/k3_p2.o /tmp/k3_p4.o /tmp/k3_p7.o /tmp/k3_p8.o /tmp/k3_p16.o /tmp/k3_p32.o /tmp/k3_p64.o /tmp/k3_p128.o /tmp/k3_tl.o -o tools/conv_ab
