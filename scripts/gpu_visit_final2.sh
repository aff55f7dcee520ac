#!/bin/bash
# final visit of round 5 (second): the full visit + the training step's kernel table
bash scripts/gpu_visit.sh r5final2
bash scripts/gpu_visit_r5n.sh > gpurun_out/r5final2/training_profile.log 2>&1
cp gpurun_out/r5n/backward_kernel_stats.csv gpurun_out/r5final2/training_step_kernel_stats.csv
tail -3 gpurun_out/r5n/backward_probe.log
