#!/bin/bash
# The gate on its own: smoke + the whole -m gpu suite three times on this HEAD, then the default bench line and the
# bf16 128^3 side workload with their per-op tables.  bash scripts/gpu_tests_x3.sh <tag>
TAG=${1:-x3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > $OUT/device.txt
nproc >> $OUT/device.txt
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log
: > $OUT/pytest_x3.txt
for pass in 1 2 3; do
  timeout 1500 python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider --durations=12 > $OUT/pytest_gpu_$pass.log 2>&1
  echo "pass $pass: pytest rc=$? | $(tail -1 $OUT/pytest_gpu_$pass.log)" | tee -a $OUT/pytest_x3.txt
  grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu_$pass.log | tee -a $OUT/pytest_x3.txt
done
grep -hE "loss.backward\(\) through|deterministic scatter|north-star perf-mode" $OUT/pytest_gpu_*.log >> $OUT/pytest_x3.txt
tail -40 $OUT/pytest_gpu_1.log
echo "== bench"; HOLO_DEBUG_PLAN=1 HOLO_BENCH_OPS=1 timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
python scripts/ops_table.py $OUT/bench.err > $OUT/ops_table.txt 2>/dev/null
B="--no-cpu-baseline --steps 6 --warmup 2 --frames 8 --flyaround-frames 0"
HOLO_BENCH_OPS=1 timeout 600 python bench.py $B --workload donut128 --compute-dtype bf16 > $OUT/donut128_bf16.json 2> $OUT/donut128_bf16.err; echo "bf16 rc=$?"
python scripts/ops_table.py $OUT/donut128_bf16.err > $OUT/donut128_bf16_ops.txt 2>/dev/null; head -80 $OUT/donut128_bf16_ops.txt
