#!/bin/bash
# Copies the judged summaries of a GPU visit from gpurun_out/<tag>/ into profiles/ under this round's names.
# bash scripts/collect_profiles.sh <tag> [round prefix, default r06_final]
T=gpurun_out/$1
P=profiles/${2:-r06_final}
cp $T/bench.json ${P}_bench.json
cp $T/kernel_stats.csv ${P}_kernel_stats.csv
cp $T/kernel_trace_by_grid.csv ${P}_kernel_trace_by_grid.csv
cp $T/ops_table.txt ${P}_ops_table.txt
cp $T/render_probe.log ${P}_render_probe.log
cp $T/backward_probe.log ${P}_backward_probe.log
cp $T/viewpool_probe.log ${P}_viewpool_probe.log
cp $T/device.txt ${P}_device.txt
cp $T/smoke.log ${P}_smoke.log
cp $T/pytest_x3.txt profiles/r06_pytest_x3.txt
cp $T/wino3_64cubed_trace.csv profiles/r06_wino3_64cubed_trace.csv
[ -f $T/conv_ab.txt ] && cp $T/conv_ab.txt profiles/r06_conv_ab.txt
tail -25 $T/pytest_gpu_3.log > ${P}_pytest_gpu_tail.txt
ls -la ${P}_* profiles/r06_pytest_x3.txt profiles/r06_wino3_64cubed_trace.csv
