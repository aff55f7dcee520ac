#!/bin/bash
# round-5 visit i: view-pooling backward with the row-segmented scatter
OUT=gpurun_out/r5i
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_viewpool.py -m gpu -x -q > $OUT/pytest_viewpool.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_viewpool.log
for cfg in "$@"; do
  name=${cfg%%:*}; envs=${cfg#*:}; nv=16
  case $name in v4*) nv=4;; esac
  env $envs timeout 300 python scripts/viewpool_probe.py $nv 64 > $OUT/viewpool_probe_$name.log 2>&1
  echo "$name: $(grep -E 'view pooling backward' $OUT/viewpool_probe_$name.log | cut -c1-90)"
done
