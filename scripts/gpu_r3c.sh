#!/bin/bash
# round-3 visit c: backward after the reduction rewrites, bf16 wide-tile timeline old/new epilogue, bench with the batched-chains line
OUT=gpurun_out/r3c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_backward.py -m gpu -q --tb=short -p no:cacheprovider -x -s 2>&1 | grep -v "^$" | tail -12 > $OUT/backward.txt; tail -8 $OUT/backward.txt
timeout 120 python scripts/backward_probe.py 5 2>&1 | tail -1
for t in old new; do for a in "128 64 64 3 0 0 1 0" "128 64 64 3 0 0 1 1" "128 128 64 3 0 0 1 1"; do echo "== $t $a"; timeout 60 ./tools/conv_timeline_$t $a | sed -n 2,4p; done; done > $OUT/timeline.txt 2>&1; cat $OUT/timeline.txt
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
b = json.load(open("gpurun_out/r3c/bench.json"))
print(b["value"], b["ms_per_step"], b.get("render", {}).get("rays_per_sec"))
print(json.dumps(b.get("side_workloads"), indent=0)[:1500])
PY
