#!/bin/bash
# per-op breakdown of one 128^3 forward in a compute mode.  bash scripts/gpu_ops128.sh <tag> <compute-dtype>
TAG=${1:-ops128}; DT=${2:-bf16}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
HOLO_BENCH_OPS=1 timeout 600 python bench.py --no-cpu-baseline --workload donut128 --compute-dtype $DT --steps 4 --warmup 2 --frames 8 --flyaround-frames 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
grep "per-op totals" $OUT/bench.err
grep "^# op" $OUT/bench.err | python3 -c "
import sys, json
for l in sys.stdin:
    o = json.loads(l[5:])
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in o.items()})
" > $OUT/ops.txt
head -c 300 $OUT/bench.json; echo
