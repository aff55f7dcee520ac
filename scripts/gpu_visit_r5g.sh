#!/bin/bash
# round-5 visit g: view-pooling backward, occupancy form (view_pool_bwd2_kernel) vs the register-accumulating one
OUT=gpurun_out/r5g
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest viewpool"
timeout 600 python -m pytest tests/test_viewpool.py -m gpu -x -q > $OUT/pytest_viewpool.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_viewpool.log
for cfg in "default:" "occ4:HOLO_VIEWPOOL_BWD_OCC=4" "v1:HOLO_VIEWPOOL_BWD_V1=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo "== probe 16 views ($name)"
  env $envs timeout 300 python scripts/viewpool_probe.py 16 64 > $OUT/viewpool_probe_$name.log 2>&1
  grep -E "view pooling" $OUT/viewpool_probe_$name.log | cut -c1-200
done
echo "== probe 4 views (default)"
timeout 300 python scripts/viewpool_probe.py 4 64 > $OUT/viewpool_probe4.log 2>&1; grep -E "view pooling backward" $OUT/viewpool_probe4.log | cut -c1-200
echo "== kernel trace of the 16-view probe"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp_prof -o vp -- python $GRAFT_REPO_ROOT/scripts/viewpool_probe.py 16 64 > /dev/null 2>&1 )
f=$(find /tmp/vp_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200 | tee $OUT/viewpool_kernel_stats.csv
