#!/bin/bash
# round-5 visit e: view-pooling backward through LDS tiles (A/B against direct atomics), MLPMean backward in voxel chunks,
# stream kernel with the weights in LDS, bf16 128^3 per-op table
OUT=gpurun_out/r5e
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest (viewpool, unet, training mode)"
timeout 1500 python -m pytest tests/test_viewpool.py tests/test_gpu_unet.py tests/test_gpu_training_mode.py tests/test_gpu_backward.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -12 $OUT/pytest_gpu.log
echo "== viewpool probe: default (LDS tiles)"
timeout 200 python scripts/viewpool_probe.py 16 64 > $OUT/viewpool_probe.log 2>&1; grep -E "view pooling|MLPMean" $OUT/viewpool_probe.log | cut -c1-220
echo "== viewpool probe: direct atomics"
HOLO_VIEWPOOL_BWD_V1=1 timeout 200 python scripts/viewpool_probe.py 16 64 > $OUT/viewpool_probe_v1.log 2>&1; grep -E "view pooling backward" $OUT/viewpool_probe_v1.log | cut -c1-220
echo "== viewpool probe: 4 views"
timeout 200 python scripts/viewpool_probe.py 4 64 > $OUT/viewpool_probe4.log 2>&1; grep -E "view pooling backward|MLPMean" $OUT/viewpool_probe4.log | cut -c1-220
echo "== bench default"
HOLO_BENCH_OPS=1 timeout 600 python bench.py --no-cpu-baseline --no-side --no-opt-in > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "import json; d=json.load(open('$OUT/bench_default.json')); print('default:', d['value'], d['ms_per_step'], d['rays_per_sec'])"
python scripts/ops_table.py $OUT/bench_default.err > $OUT/ops_table.txt; grep -E "x1_st|sum" $OUT/ops_table.txt
echo "== bf16 128^3 per-op table"
HOLO_BENCH_OPS=1 timeout 600 python bench.py --workload donut128 --compute-dtype bf16 --steps 5 --warmup 3 --frames 2 --flyaround-frames 0 --no-cpu-baseline --no-side --no-opt-in > $OUT/bench_donut_bf16.json 2> $OUT/bench_donut_bf16.err
python -c "import json; d=json.load(open('$OUT/bench_donut_bf16.json')); print('donut128 bf16:', d['value'], d['ms_per_step'])"
python scripts/ops_table.py $OUT/bench_donut_bf16.err > $OUT/ops_donut_bf16.txt; cat $OUT/ops_donut_bf16.txt | head -70
echo "== row-tile kernel: split-K target 1 / 3 / 4 workgroups per CU"
for t in 1 3 4; do
  HOLO_SMALL_SPLIT_TARGET=$t HOLO_BENCH_OPS=1 timeout 600 python bench.py --no-cpu-baseline --no-side --no-opt-in --frames 2 --flyaround-frames 0 > $OUT/bench_split$t.json 2> $OUT/bench_split$t.err
  python -c "import json; d=json.load(open('$OUT/bench_split$t.json')); print('split target $t:', d['value'], d['ms_per_step'])"
  grep "per-op totals" $OUT/bench_split$t.err | cut -c1-300
done
