#!/bin/bash
# round-5 visit f: bf16 128^3 per-op table, MLPMean backward with 65 536-voxel chunks, bench line with the normals rate
OUT=gpurun_out/r5f
mkdir -p $OUT
export TMPDIR=/tmp
echo "== viewpool probe (4 views)"
timeout 200 python scripts/viewpool_probe.py 4 64 > $OUT/viewpool_probe4.log 2>&1; grep -E "view pooling backward|MLPMean" $OUT/viewpool_probe4.log | cut -c1-220
echo "== bf16 128^3 per-op table"
HOLO_DEBUG_PLAN=1 HOLO_BENCH_OPS=1 timeout 600 python bench.py --workload donut128 --compute-dtype bf16 --steps 5 --warmup 3 --frames 2 --flyaround-frames 0 --no-cpu-baseline --no-side --no-opt-in > $OUT/bench_donut_bf16.json 2> $OUT/bench_donut_bf16.err
python -c "import json; d=json.load(open('$OUT/bench_donut_bf16.json')); print('donut128 bf16:', d['value'], d['ms_per_step'])"
python scripts/ops_table.py $OUT/bench_donut_bf16.err > $OUT/ops_donut_bf16.txt; cat $OUT/ops_donut_bf16.txt | head -80
echo "== bench (short)"
timeout 600 python bench.py --no-cpu-baseline --no-side --no-opt-in > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "import json; d=json.load(open('$OUT/bench_default.json')); print('default:', d['value'], d['ms_per_step'], d['rays_per_sec'], d['rays_per_sec_with_normals'])"
