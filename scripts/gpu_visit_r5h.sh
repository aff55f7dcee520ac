#!/bin/bash
# round-5 visit h: where the time of the view-pooling backward goes (development probes of view_pool_bwd2_kernel)
# usage: gpu_visit_r5h.sh name:ENV=val ...
OUT=gpurun_out/r5h
mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "$@"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs HOLO_VIEWPOOL_BWD_OCC=4 timeout 300 python scripts/viewpool_probe.py 16 64 > $OUT/viewpool_probe_$name.log 2>&1
  echo "$name: $(grep -E 'view pooling backward' $OUT/viewpool_probe_$name.log | cut -c1-90)"
done
