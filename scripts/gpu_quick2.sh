#!/bin/bash
# quick GPU visit: selected tests + short bench (no CPU baseline).  bash scripts/gpu_quick2.sh <tag> "<pytest -k expr>" [extra bench args]
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "$2" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest.log
HOLO_BENCH_OPS=1 timeout 600 python bench.py --no-cpu-baseline $3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python3 - <<PY
import json
d=json.load(open("$OUT/bench.json")); r=d["roofline"]
print("steps/s %.2f ms/step %.3f | rays/s %.4g ms/frame %.3f"%(d["value"],d["ms_per_step"],d["rays_per_sec"],d["ms_per_frame"]))
print("dominant:", r["kernel"][:80], "eff TF %.1f exec TF %.1f frac %.3f avg_ms %.4f"%(r["effective_tflops"],r["achieved"],r["frac"],r["avg_launch_ms"]))
for v in r["by_variant"]: print("  %-20s wc%d tz%d sk%d od%-3d n%-2d ms %.3f TF %.1f exec %.1f"%(v["kernel"],v["wave_cols"],v["tile_depth"],v["fused_skip"],v["out_dim"],v["launches"],v["ms"],v["tflops"],v["tflops_executed"]))
print("opt-in:", {k:(round(v["denoise_steps_per_s"],1), round(v.get("rays_per_sec",0)/1e6,1)) for k,v in (d.get("opt_in_modes_not_reported") or {}).items()})
PY
