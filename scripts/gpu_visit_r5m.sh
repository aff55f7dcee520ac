#!/bin/bash
# round-5 visit m: GroupNorm finalize fused into the split-K reduce (A/B by HOLO_NO_FINAL_FUSION) + the UNet-side tests
OUT=gpurun_out/r5m
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest (unet, backward, diffusion, training mode, configs)"
timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_backward.py tests/test_gpu_diffusion.py tests/test_gpu_training_mode.py tests/test_gpu_configs.py -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
for cfg in "fused:" "unfused:HOLO_NO_FINAL_FUSION=1" "fused2:"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs HOLO_DEBUG_PLAN=1 HOLO_BENCH_OPS=1 timeout 600 python bench.py --steps 40 --warmup 10 --frames 2 --flyaround-frames 0 --no-cpu-baseline --no-side --no-opt-in > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "import json; d=json.load(open('$OUT/bench_$name.json')); print('$name:', d['value'], d['ms_per_step'])"
  grep "\[plan\] batch" $OUT/bench_$name.err | tail -1
done
python scripts/ops_table.py $OUT/bench_unfused.err $OUT/bench_fused.err > $OUT/ops_compare.txt 2>&1; tail -5 $OUT/ops_compare.txt
