#!/bin/bash
# One GPU-box visit: smoke, -m gpu parity tests (three passes), render probes, bench line, rocprofv3 trace of the EXACT
# bench command.  bash scripts/gpu_visit.sh <tag> [pytest -k expression]
TAG=${1:-v}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > $OUT/device.txt
nproc >> $OUT/device.txt; lscpu | grep -E "Model name|Socket|Core|Thread" >> $OUT/device.txt
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log
echo "== pytest -m gpu, THREE times on this HEAD (no -x: every failure of every pass is listed) -> pytest_x3.txt"
git -C . rev-parse --short HEAD > /dev/null 2>&1 || true
: > $OUT/pytest_x3.txt
for pass in 1 2 3; do
  if [ -n "$2" ]; then
    timeout 1500 python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider --durations=12 -k "$2" > $OUT/pytest_gpu_$pass.log 2>&1
  else
    timeout 1500 python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider --durations=12 > $OUT/pytest_gpu_$pass.log 2>&1
  fi
  echo "pass $pass: pytest rc=$? | $(tail -1 $OUT/pytest_gpu_$pass.log)" | tee -a $OUT/pytest_x3.txt
  grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu_$pass.log | tee -a $OUT/pytest_x3.txt
done
grep -hE "loss.backward\(\) through|deterministic scatter|north-star perf-mode" $OUT/pytest_gpu_*.log >> $OUT/pytest_x3.txt
tail -30 $OUT/pytest_gpu_1.log
echo "== render probes"
timeout 300 python scripts/render_probe.py 1 8 30 > $OUT/render_probe.log 2>&1
HOLO_RENDER_XCD=1 timeout 300 python scripts/render_probe.py 8 >> $OUT/render_probe.log 2>&1
HOLO_RENDER_TIMELINE=1 timeout 300 python scripts/render_probe.py 8 2>&1 | grep -m2 "timeline" >> $OUT/render_probe.log
cat $OUT/render_probe.log
echo "== probes: training step (forward + backward), view pooling forward / backward"
timeout 200 python scripts/backward_probe.py 5 > $OUT/backward_probe.log 2>&1; tail -2 $OUT/backward_probe.log
timeout 300 python scripts/viewpool_probe.py 16 64 > $OUT/viewpool_probe.log 2>&1; grep -E "view pooling|MLPMean" $OUT/viewpool_probe.log
echo "== bench"; HOLO_DEBUG_PLAN=1 HOLO_BENCH_OPS=1 timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; grep -E "per-op totals|\[plan\]" $OUT/bench.err | head -3
python scripts/ops_table.py $OUT/bench.err > $OUT/ops_table.txt 2>/dev/null
echo "== rocprofv3 kernel trace of the bench command"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err; echo "rocprof rc=$?" )
for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
for f in $(find /tmp/prof_$TAG -name "*kernel_trace.csv"); do
  python3 scripts/trace_by_grid.py "$f" "$OUT/kernel_trace_by_grid.csv"
  # conv_wino3_kernel per level (every level launches the same grid): the trace joined with the plan's op order
  python3 scripts/wino3_trace_join.py "$f" $OUT/bench.err 60 $OUT/wino3_64cubed_trace.csv
done
head -14 $OUT/kernel_stats.csv 2>/dev/null | cut -c1-220
