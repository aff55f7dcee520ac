#!/bin/bash
# round-3 visit d: renderer backward + chain, training-mode forward (the kernel now emits the merged list), bf16 tests after the epilogue change
OUT=gpurun_out/r3d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_render_backward.py tests/test_gpu_training_mode.py -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -v "^$" | tail -40 > $OUT/render_bwd.txt; tail -30 $OUT/render_bwd.txt
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_diffusion.py -m gpu -q --tb=short -p no:cacheprovider -k "bf16" 2>&1 | tail -8 > $OUT/bf16.txt; tail -8 $OUT/bf16.txt
