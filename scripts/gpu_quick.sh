#!/bin/bash
# quick GPU visit: selected tests + bench (no cpu baseline) + by-grid kernel trace.  bash scripts/gpu_quick.sh tag ["-k expr"]
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider $2 > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python3 -c "
import json;d=json.load(open('$OUT/bench.json'));r=d['roofline']
print('steps/s %.2f ms/step %.2f rays/s %.3g ms/frame %.2f conv TF %.1f frac %.3f conv_ms %.2f'%(d['value'],d['ms_per_step'],d['rays_per_sec'],d['ms_per_frame'],r['achieved'],r['frac'],r['ms_per_forward']))"
bash scripts/gpu_prof.sh $TAG > /dev/null 2>&1
cut -c1-150 $OUT/kernel_trace_summary.csv | grep -v "at::native" | head -${3:-14}
