#!/bin/bash
# round-5 visits i-l: view-pooling backward kernels (tests + scripts/viewpool_probe.py under the given knobs)
# usage: gpu_visit_r5i.sh name:ENV=val ...   (a name starting with v4 probes 4 views instead of 16)
OUT=gpurun_out/r5i
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_viewpool.py -m gpu -x -q > $OUT/pytest_viewpool.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_viewpool.log
for cfg in "$@"; do
  name=${cfg%%:*}; envs=${cfg#*:}; nv=16
  case $name in v4*) nv=4;; esac
  env $envs timeout 300 python scripts/viewpool_probe.py $nv 64 > $OUT/viewpool_probe_$name.log 2>&1
  echo "$name: $(grep -E 'view pooling backward' $OUT/viewpool_probe_$name.log | cut -c1-90)"
  echo "$name: $(grep -E 'MLPMean' $OUT/viewpool_probe_$name.log | cut -c1-170)"
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp_prof -o vp -- python $GRAFT_REPO_ROOT/scripts/viewpool_probe.py 4 64 > /dev/null 2>&1 )
f=$(find /tmp/vp_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-160 | tee $OUT/viewpool_kernel_stats.csv
