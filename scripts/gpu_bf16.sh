#!/bin/bash
# bf16 storage mode: parity tests (wide-tile kernel forced / planner's choice) + per-op breakdown at 128^3 + kernel timeline
TAG=${1:-bf16}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
HOLO_CONV_BF16T=1 timeout 600 python -m pytest tests/test_gpu_unet.py -m gpu -q --tb=short -p no:cacheprovider -x -k "bf16_compute_mode_vs_oracle or bf16_flash" > $OUT/pytest_forced.log 2>&1; echo "forced rc=$?"; tail -5 $OUT/pytest_forced.log
timeout 900 python -m pytest tests/test_gpu_unet.py -m gpu -q --tb=short -p no:cacheprovider -k "bf16 or 128_cubed" > $OUT/pytest_unet.log 2>&1; echo "unet rc=$?"; tail -5 $OUT/pytest_unet.log
bash scripts/gpu_ops128.sh $TAG/ops bf16
for a in "128 64 64 3 0 0 0" "128 64 64 3 0 0 1" "128 128 64 3 0 0 1"; do echo "== $a"; ./tools/conv_timeline $a | head -5; done
