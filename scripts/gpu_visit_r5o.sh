#!/bin/bash
# round-5 visit o: backward tests + the training step's time (scripts/backward_probe.py), optionally under knobs: name:ENV=val ...
OUT=gpurun_out/r5o
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_training_mode.py -m gpu -x -q > $OUT/pytest_bwd.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_bwd.log
for cfg in "$@"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python scripts/backward_probe.py 10 > $OUT/backward_probe_$name.log 2>&1
  echo "$name: $(tail -1 $OUT/backward_probe_$name.log)"
done
