#!/bin/bash
# Round-3 GPU visit: bash scripts/gpu_r3.sh <tag> [pytest -k expr] [bench args]
TAG=${1:-r3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
if [ -n "$2" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider --durations=8 -k "$2" > $OUT/pytest_gpu.log 2>&1
else
  timeout 1500 python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider --durations=8 > $OUT/pytest_gpu.log 2>&1
fi
echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "drift|depth control|passed|failed|FAILED|Error|error" $OUT/pytest_gpu.log | tail -40
if [ "$3" != "nobench" ]; then
echo "== bench"; HOLO_BENCH_OPS=1 timeout 900 python bench.py $3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -4 $OUT/bench.err | cut -c1-1500
python3 - <<PY
import json
d=json.load(open("$OUT/bench.json")); r=d["roofline"]
print("steps/s %.2f ms/step %.3f | rays/s %.4g ms/frame %.3f | second %.4g"%(d["value"],d["ms_per_step"],d["rays_per_sec"],d["ms_per_frame"],d.get("rays_per_sec_second_call_size") or 0))
print("dominant:", r["kernel"][:70], "eff TF %.1f exec TF %.1f frac %.3f avg_ms %.4f"%(r["effective_tflops"],r["achieved"],r["frac"],r["avg_launch_ms"]))
print("side:", d.get("side_workloads")); print("cpu:", {k:v for k,v in (d.get("cpu_baseline") or {}).items() if k!="sample"})
PY
fi
