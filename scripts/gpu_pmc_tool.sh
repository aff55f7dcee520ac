#!/bin/bash
# PMC counters of a standalone probe binary, per kernel: bash scripts/gpu_pmc_tool.sh <tag> <command...>
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="$@"
run_pass () {
  name=$1; shift
  ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$name -o p -- $CMD > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_$name.err ; echo "pass $name rc=$?" )
  for f in $(find /tmp/pmc_${TAG}_$name -name "*counter_collection.csv"); do
    python3 - "$f" "$OUT/pmc_$name.csv" <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:70], r.get("Grid_Size", r.get("Grid_Size_X", "")), r["Counter_Name"])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
with open(sys.argv[2], "w") as f:
    f.write("kernel,grid,counter,dispatches,sum,avg\n")
    for k, a in agg.items():
        f.write(f"\"{k[0]}\",{k[1]},{k[2]},{a[0]},{a[1]:.0f},{a[1]/a[0]:.1f}\n")
PY
  done
}
run_pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run_pass grbm GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU
grep -h "flash_attn_bf16v2" $OUT/pmc_sq.csv $OUT/pmc_grbm.csv | cut -c40-200
