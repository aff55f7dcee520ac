#!/usr/bin/env python3
"""rocprofv3 kernel_trace.csv -> per (kernel, grid) summary: calls, total/avg microseconds, VGPRs, LDS."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"][:110], r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(k, [0, 0, r.get("VGPR_Count", ""), r.get("LDS_Block_Size", "")])
    a[0] += 1
    a[1] += d
with open(sys.argv[2], "w") as f:
    f.write("kernel,grid_x,grid_y,grid_z,calls,total_us,avg_us,vgpr,lds\n")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f"\"{k[0]}\",{k[1]},{k[2]},{k[3]},{a[0]},{a[1]/1e3:.1f},{a[1]/1e3/a[0]:.2f},{a[2]},{a[3]}\n")
