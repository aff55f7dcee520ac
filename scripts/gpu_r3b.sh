#!/bin/bash
# render A/B + selected tests: bash scripts/gpu_r3b.sh <tag> "<pytest -k>"
TAG=${1:-r3b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider -k "$2" > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "drift|PER-STEP|depth control|bad ray|passed|failed|FAILED|^E  " $OUT/pytest_gpu.log | cut -c1-900 | tail -40
echo "== render probe: new kernel"; timeout 300 python scripts/render_probe.py 1 8 40 2>&1 | grep "normals 0" | tee $OUT/probe_new.txt
echo "== render probe: new kernel, 8 waves per CU (HOLO_RENDER2_NW=8)"; HOLO_RENDER2_NW=8 timeout 300 python scripts/render_probe.py 1 8 40 2>&1 | grep "normals 0" | tee $OUT/probe_new_nw8.txt
echo "== render probe: ray-per-column kernel (HOLO_RENDER_V1=1)"; HOLO_RENDER_V1=1 timeout 300 python scripts/render_probe.py 1 8 40 2>&1 | grep "normals 0" | tee $OUT/probe_v1.txt
