#!/bin/bash
# PMC counters of one conv_timeline configuration: bash scripts/gpu_pmc_tool.sh <tag> "<conv_timeline args>"
TAG=${1:-pmct}; ARGS=$2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run_pass () {
  name=$1; shift
  ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$name -o p -- $GRAFT_REPO_ROOT/tools/conv_timeline $ARGS > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_$name.err ; echo "pass $name rc=$?" )
  for f in $(find /tmp/pmc_${TAG}_$name -name "*counter_collection.csv"); do
    python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:60], r["Counter_Name"])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for k, a in agg.items(): print(k[0], k[1], a[0], "%.0f" % (a[1] / a[0]))
PY
  done
}
run_pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run_pass grbm GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
