#!/bin/bash
# A shorter GPU-box visit: -m gpu parity tests, render probe, bench line with the per-op table (no rocprof pass).
# bash scripts/gpu_visit_lite.sh <tag> [pytest -k expression]
TAG=${1:-v}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > $OUT/device.txt
echo "== pytest -m gpu"
if [ -n "$2" ]; then
  timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=15 -k "$2" > $OUT/pytest_gpu.log 2>&1
else
  timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=15 > $OUT/pytest_gpu.log 2>&1
fi
echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -60 $OUT/pytest_gpu.log
echo "== render probes"
timeout 300 python scripts/render_probe.py 1 8 30 > $OUT/render_probe.log 2>&1; cat $OUT/render_probe.log
echo "== bench"; HOLO_DEBUG_PLAN=1 HOLO_BENCH_OPS=1 timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; grep -E "per-op totals" $OUT/bench.err | head -3
python scripts/ops_table.py $OUT/bench.err > $OUT/ops_table.txt 2>/dev/null; head -60 $OUT/ops_table.txt
