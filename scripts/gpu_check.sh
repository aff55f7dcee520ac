#!/bin/bash
# One GPU-box visit: smoke, -m gpu parity tests, bench line, rocprofv3 kernel-trace summary.
# Usage (from the repo root, on the GPU box via gpurun):  bash scripts/gpu_check.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > $OUT/device.txt
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 ; echo "smoke rc=$?" | tee -a $OUT/smoke.log
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
echo "== bench" ; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ; echo "bench rc=$?" ; cat $OUT/bench.json ; tail -5 $OUT/bench.err
echo "== rocprofv3 kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --frames 2 --no-cpu-baseline --conv-iters 1 > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err ; echo "rocprof rc=$?" )
find /tmp/prof_$TAG -name "*stats*" | head ; 
for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
for f in $(find /tmp/prof_$TAG -name "*kernel_trace.csv"); do python3 scripts/trace_by_grid.py "$f" "$OUT/kernel_trace_by_grid.csv"; done
head -12 $OUT/kernel_stats.csv 2>/dev/null | cut -c1-200
