#!/bin/bash
# rocprofv3 kernel-trace summary of a short bench run (+ optional pytest selection).  bash scripts/gpu_prof.sh tag [pytest -k expr]
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -n "$2" ]; then timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "$2" > $OUT/pytest_sel.log 2>&1; tail -5 $OUT/pytest_sel.log; fi
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --frames 2 --no-cpu-baseline --conv-iters 1 > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err ; echo "rocprof rc=$?" )
find /tmp/prof_$TAG -type f | head
for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
for f in $(find /tmp/prof_$TAG -name "*kernel_trace.csv"); do python3 - "$f" "$OUT/kernel_trace_summary.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"][:90], r.get("Grid_Size_X", ""), r.get("Workgroup_Size_X",""))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(k, [0, 0, r.get("VGPR_Count",""), r.get("LDS_Block_Size","")])
    a[0] += 1; a[1] += d
with open(sys.argv[2], "w") as f:
    f.write("kernel,grid_x,wg_x,calls,total_us,avg_us,vgpr,lds\n")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f"\"{k[0]}\",{k[1]},{k[2]},{a[0]},{a[1]/1e3:.1f},{a[1]/1e3/a[0]:.2f},{a[2]},{a[3]}\n")
PY
done
head -40 $OUT/kernel_stats.csv
