#!/usr/bin/env python3
"""Per-level times of conv_wino3_kernel from a rocprofv3 kernel trace.

Every launch of the persistent kernel has the same grid (one workgroup per CU), so `--stats` mixes the 64^3 launches with
the 32^3 / 16^3 / 8^3 ones.  This joins the trace with the plan instead: the denoise leg of `bench.py` is the first thing
that launches the kernel, each forward launches the plan's conv_wino3 ops in plan order, and `HOLO_BENCH_OPS=1` dumps that
order (`# op` lines of bench.err: out_dim, channels, fused skip per op).  Dispatch i of the kernel (in start-time order)
therefore belongs to op i mod n of the plan, for the first (warmup + steps) * n dispatches.

    wino3_trace_join.py <kernel_trace.csv> <bench.err with '# op' lines> <warmup + steps> <out.csv>
"""
import collections
import csv
import json
import sys

trace, err, forwards, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
ops = [json.loads(l[5:]) for l in open(err) if l.startswith("# op")]
w3 = [o for o in ops if o["op"] == "conv" and o.get("kernel") == "conv_wino3_kernel"]
rows = [r for r in csv.DictReader(open(trace)) if "conv_wino3_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(w3)
use = rows[: forwards * n]
if n == 0 or len(use) < forwards * n:
    sys.exit(f"plan has {n} conv_wino3 ops, trace has {len(rows)} dispatches: cannot join {forwards} forwards")
per_op = collections.defaultdict(list)
for i, r in enumerate(use):
    fused = "true" in r["Kernel_Name"].split("conv_wino3_kernel<")[1].split(",")[0]
    o = w3[i % n]
    if fused != bool(o["fused_skip"]):  # the template argument in the kernel name is an independent check of the join
        sys.exit(f"dispatch {i}: kernel name {r['Kernel_Name'][:80]} does not match plan op {i % n} (fused_skip {o['fused_skip']})")
    per_op[i % n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
levels = collections.OrderedDict()
with open(out, "w") as f:
    f.write("plan_position,out_dim,cin,cout,fused_skip,nsplit,dispatches,avg_us,min_us,max_us,bench_hipevent_us\n")
    for j, o in enumerate(w3):
        d = per_op[j][forwards // 6:]  # (the first forwards are warm-up: clocks and caches settle)
        f.write(f"{j},{o['out_dim']},{o['cin']},{o['cout']},{int(o['fused_skip'])},{o['nsplit']},{len(d)},"
                f"{sum(d) / len(d):.2f},{min(d):.2f},{max(d):.2f},{o['ms'] * 1e3:.2f}\n")
        k = (o["out_dim"], bool(o["fused_skip"]))
        a = levels.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += sum(d) / len(d)
        a[2] += o["ms"] * 1e3
    f.write("# per level: out_dim,fused_skip,launches_per_forward,avg_us_per_launch (rocprofv3 trace),avg_us_per_launch (bench.py hipEvents)\n")
    for (od, fs), a in levels.items():
        line = f"# level,{od},{int(fs)},{a[0]},{a[1] / a[0]:.2f},{a[2] / a[0]:.2f}"
        f.write(line + "\n")
        print(line)
