#!/bin/bash
# Side workloads (never the reported line): 128^3x32 in fp32 and in the bf16 mode (per-op table + rocprofv3 kernel stats
# of the bf16 run), 32^3x16.  bash scripts/gpu_side.sh <tag>
TAG=${1:-side}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --steps 6 --warmup 2 --frames 8 --flyaround-frames 0"
HOLO_BENCH_OPS=1 timeout 600 python bench.py $B --workload donut128 --compute-dtype bf16 > $OUT/donut128_bf16.json 2> $OUT/donut128_bf16.err; echo "bf16 rc=$?"
grep "per-op totals" $OUT/donut128_bf16.err
timeout 600 python bench.py $B --workload donut128 > $OUT/donut128.json 2> $OUT/donut128.err; echo "f32 rc=$?"
timeout 600 python bench.py $B --workload small > $OUT/small.json 2> $OUT/small.err; echo "small rc=$?"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python $GRAFT_REPO_ROOT/bench.py $B --workload donut128 --compute-dtype bf16 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err; echo "rocprof rc=$?" )
for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv"); do cp $f $OUT/donut128_bf16_kernel_stats.csv; done
head -12 $OUT/donut128_bf16_kernel_stats.csv | cut -c1-200
python3 - <<PY
import json
for n in ("donut128_bf16", "donut128", "small"):
    d = json.load(open("$OUT/%s.json" % n)); r = d["roofline"]
    print(n, "steps/s %.2f ms %.3f ws %.0f MB | dominant %s: %.0f TF frac %.3f avg %.4f ms" % (d["value"], d["ms_per_step"], d["unet_workspace_bytes"] / 1e6, r["kernel"][:48], r["achieved"], r["frac"], r["avg_launch_ms"]))
PY
