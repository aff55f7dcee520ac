#!/bin/bash
# bf16 128^3 (BASELINE configs[4]) A/B of the wide-tile convolution kernel's two forms + the bf16 parity tests.
# bash scripts/gpu_bf16_ab.sh <tag>
TAG=${1:-bf16ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== bf16 parity tests (the persistent form is the default on the filled levels)"
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_diffusion.py -m gpu -q -s --tb=short -p no:cacheprovider -k "bf16 or 128_cubed or donut" > $OUT/pytest_bf16.log 2>&1
echo "pytest rc=$? | $(tail -1 $OUT/pytest_bf16.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_bf16.log
B="--no-cpu-baseline --steps 6 --warmup 2 --frames 8 --flyaround-frames 0"
for mode in 0 on 2; do
  if [ "$mode" = "on" ]; then unset HOLO_CONV_BF16P; else export HOLO_CONV_BF16P=$mode; fi
  HOLO_DEBUG_PLAN=1 HOLO_BENCH_OPS=1 timeout 600 python bench.py $B --workload donut128 --compute-dtype bf16 > $OUT/donut128_bf16_p$mode.json 2> $OUT/donut128_bf16_p$mode.err
  echo "bf16 p=$mode rc=$?"; grep "per-op totals" $OUT/donut128_bf16_p$mode.err | cut -c1-400
done
unset HOLO_CONV_BF16P
python scripts/ops_table.py $OUT/donut128_bf16_p0.err $OUT/donut128_bf16_pon.err $OUT/donut128_bf16_p2.err > $OUT/donut128_bf16_ab_ops.txt 2>/dev/null; cat $OUT/donut128_bf16_ab_ops.txt
python3 - <<PY
import json
for n in ("p0", "pon", "p2"):
    d = json.load(open("$OUT/donut128_bf16_%s.json" % n)); r = d["roofline"]
    print(n, "steps/s %.2f ms %.3f | dominant %s: %.0f TF frac %.3f avg %.4f ms" % (d["value"], d["ms_per_step"], r["kernel"][:60], r["achieved"], r["frac"], r["avg_launch_ms"]))
PY
