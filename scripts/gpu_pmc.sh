#!/bin/bash
# PMC counter passes (separate from tracing, as the pool requires): bash scripts/gpu_pmc.sh tag [traffic]   (traffic: only FETCH_SIZE / WRITE_SIZE)
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# every render dispatch of this command is a 20-frame launch (what a 40-frame call splits into) (frames_per_launch of scripts/pmc_to_json.py); no opt-in modes
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --frames 20 --flyaround-frames 0 --no-opt-in --no-cpu-baseline --no-side --conv-iters 1"
run_pass () {
  name=$1; shift
  ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$name -o p -- $CMD > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_$name.err ; echo "pass $name rc=$?" )
  for f in $(find /tmp/pmc_${TAG}_$name -name "*counter_collection.csv"); do
    python3 - "$f" "$OUT/pmc_$name.csv" <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    grid = r.get("Grid_Size", r.get("Grid_Size_X", ""))
    # conv_wino3_kernel launches 256 persistent workgroups on EVERY level: its 64^3 launches (the reported ones) are told from
    # the rest by the traffic itself (>= 48 MiB on the raw counter: a 64^3 launch fetches >= 90 MiB raw and writes 67 MiB, the fused-skip launches of the 32^3 level reach 28 MiB)
    if "conv_wino3_kernel" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE") and float(r["Counter_Value"]) >= 49152:
        grid += "@64"
    k = (r["Kernel_Name"][:110], grid, r["Counter_Name"])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += float(r["Counter_Value"])
with open(sys.argv[2], "w") as f:
    f.write("kernel,grid,counter,dispatches,sum,avg\n")
    for k, a in agg.items():
        if "at::native" in k[0] or "rocclr" in k[0]: continue
        f.write(f"\"{k[0]}\",{k[1]},{k[2]},{a[0]},{a[1]:.0f},{a[1]/a[0]:.1f}\n")
PY
  done
}
if [ "$2" != "traffic" ]; then
# (round 5: the eight SQ counters in ONE pass aborted the queue - HSA_STATUS_ERROR_INVALID_PACKET_FORMAT six seconds into the
#  run, then sat out the timeout; two passes of four counters run clean)
run_pass sqa SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY
run_pass sqb SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
if [ -f $OUT/pmc_sqa.csv ]; then cp $OUT/pmc_sqa.csv $OUT/pmc_sq.csv; [ -f $OUT/pmc_sqb.csv ] && tail -n +2 $OUT/pmc_sqb.csv >> $OUT/pmc_sq.csv; fi
run_pass grbm GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
fi
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
ls -la $OUT
# afterwards, in the development container: python scripts/pmc_to_json.py <tag> profiles/pmc_traffic.json 20
#   (20 = frames per render launch of the command above) and copy gpurun_out/<tag>/pmc_*.csv to profiles/<round>_pmc_*.csv
