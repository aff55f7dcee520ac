#!/bin/bash
# round-5 visit n: kernel table of the north-star training step (forward + backward of the denoiser)
OUT=gpurun_out/r5n
mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bw_prof -o bw -- python $GRAFT_REPO_ROOT/scripts/backward_probe.py 5 > $GRAFT_REPO_ROOT/$OUT/backward_probe.log 2>&1 )
cat $OUT/backward_probe.log | tail -2
f=$(find /tmp/bw_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-230 > $OUT/backward_kernel_stats.csv
t=$(find /tmp/bw_prof -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python scripts/trace_by_grid.py "$t" > $OUT/backward_trace_by_grid.csv 2>/dev/null
head -40 $OUT/backward_kernel_stats.csv | cut -c1-200
