#!/bin/bash
# round-5 visit b: render probes (normals on render2_kernel: 8 vs 10 waves, tail items on/off), skip-fusion A/B, tests of the
# render / model / config files
OUT=gpurun_out/r5b
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest (render, model, configs, distributed noise)"
timeout 1200 python -m pytest tests/test_gpu_render.py tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_training_mode.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -15 $OUT/pytest_gpu.log
echo "== render probe: default (tail items on, normals 8 waves)"
timeout 300 python scripts/render_probe.py 1 8 30 > $OUT/render_probe_default.log 2>&1; cat $OUT/render_probe_default.log
echo "== render probe: HOLO_RENDER_TAIL=0"
HOLO_RENDER_TAIL=0 timeout 300 python scripts/render_probe.py 1 8 > $OUT/render_probe_tail0.log 2>&1; cat $OUT/render_probe_tail0.log
echo "== render probe: HOLO_RENDER_TAIL=96 / 384 / 768"
for t in 96 384 768; do HOLO_RENDER_TAIL=$t timeout 300 python scripts/render_probe.py 1 > $OUT/render_probe_tail$t.log 2>&1; echo "tail $t"; cat $OUT/render_probe_tail$t.log; done
echo "== render probe: normals on 10 waves"
HOLO_RENDER2_NRM_NW=10 timeout 300 python scripts/render_probe.py 1 8 > $OUT/render_probe_nrm10.log 2>&1; cat $OUT/render_probe_nrm10.log
echo "== render probe: old kernel (HOLO_RENDER_V1)"
HOLO_RENDER_V1=1 timeout 300 python scripts/render_probe.py 8 > $OUT/render_probe_v1.log 2>&1; cat $OUT/render_probe_v1.log
echo "== bench: skip fusion A/B"
for r in 64 32; do
  HOLO_SKIP_FUSION_BELOW_R=$r HOLO_BENCH_OPS=1 timeout 600 python bench.py --no-cpu-baseline --no-side --no-opt-in > $OUT/bench_skipR$r.json 2> $OUT/bench_skipR$r.err
  python -c "import json; d=json.load(open('$OUT/bench_skipR$r.json')); print('skip fusion below R=$r:', d['value'], d['ms_per_step'])"
  python scripts/ops_table.py $OUT/bench_skipR$r.err > $OUT/ops_skipR$r.txt
done
HOLO_BENCH_OPS=1 timeout 600 python bench.py --no-cpu-baseline --no-side --no-opt-in > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "import json; d=json.load(open('$OUT/bench_default.json')); print('default:', d['value'], d['ms_per_step'], d['rays_per_sec'])"
python scripts/ops_table.py $OUT/bench_default.err $OUT/bench_skipR64.err > $OUT/ops_compare.txt; grep -E "od64|od32|sum" $OUT/ops_compare.txt
