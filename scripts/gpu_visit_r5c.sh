#!/bin/bash
# round-5 visit c: streaming 1x1x1 skip kernel, normals from the density scalar field (8 vs 10 waves)
OUT=gpurun_out/r5c
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest (unet, render, configs)"
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_render.py tests/test_gpu_configs.py tests/test_gpu_diffusion.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -12 $OUT/pytest_gpu.log
echo "== render probe: default"
timeout 300 python scripts/render_probe.py 1 8 30 > $OUT/render_probe_default.log 2>&1; cat $OUT/render_probe_default.log
echo "== render probe: normals on 10 waves"
HOLO_RENDER2_NRM_NW=10 timeout 300 python scripts/render_probe.py 1 8 > $OUT/render_probe_nrm10.log 2>&1; grep "normals 1" $OUT/render_probe_nrm10.log
echo "== bench default"
HOLO_BENCH_OPS=1 timeout 600 python bench.py --no-cpu-baseline --no-side --no-opt-in > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "import json; d=json.load(open('$OUT/bench_default.json')); print('default:', d['value'], d['ms_per_step'], d['rays_per_sec'])"
echo "== bench with the fused skip everywhere"
HOLO_SKIP_FUSION_BELOW_R=1000 HOLO_BENCH_OPS=1 timeout 600 python bench.py --no-cpu-baseline --no-side --no-opt-in > $OUT/bench_fused.json 2> $OUT/bench_fused.err
python -c "import json; d=json.load(open('$OUT/bench_fused.json')); print('fused:', d['value'], d['ms_per_step'])"
python scripts/ops_table.py $OUT/bench_default.err $OUT/bench_fused.err > $OUT/ops_compare.txt; grep -E "od64|sum" $OUT/ops_compare.txt
