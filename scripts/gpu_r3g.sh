#!/bin/bash
# round-3 visit g: row-staged wgrad (parity + timing, A/B against the tile kernel), render backward large cases
OUT=gpurun_out/r3g
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_backward.py -m gpu -q --tb=short -p no:cacheprovider -x -s 2>&1 | grep -v "^$" | tail -14 > $OUT/backward.txt; tail -10 $OUT/backward.txt
echo "rows kernel:"; timeout 120 python scripts/backward_probe.py 5 2>&1 | tail -1
echo "tile kernel:"; HOLO_WGRAD_TILES=1 timeout 120 python scripts/backward_probe.py 5 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_render_backward.py -m gpu -q --tb=short -p no:cacheprovider -s -k "chunks or 300" 2>&1 | grep -v "^$" | tail -12 | cut -c1-600
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g2 -o prof -- python $GRAFT_REPO_ROOT/scripts/backward_probe.py 3 > /dev/null 2>&1 )
for f in $(find /tmp/prof_g2 -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
head -12 $OUT/kernel_stats.csv | cut -c1-170
