#!/usr/bin/env python3
"""rocprofv3 kernel_trace.csv of a denoise loop -> per DDPM step: wall (first start .. last end), sum of kernel durations, idle
gaps between consecutive kernels, and the kernels by total time inside the steps.  Steps are delimited by ddpm_step_kernel.
usage: step_gaps.py <kernel_trace.csv> [skip_steps=3]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps, cur = [], []
for r in rows:
    cur.append(r)
    if "ddpm_step_kernel" in r["Kernel_Name"]:
        steps.append(cur)
        cur = []
steps = steps[skip:]
tot = collections.Counter()
cnt = collections.Counter()
walls, sums, gaps, ngap = [], [], [], []
for st in steps:
    s0, e1 = int(st[0]["Start_Timestamp"]), int(st[-1]["End_Timestamp"])
    walls.append((e1 - s0) / 1e3)
    sums.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in st) / 1e3)
    g = [max(0, int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) for a, b in zip(st[:-1], st[1:])]
    gaps.append(sum(g) / 1e3)
    ngap.append(len(st))
    for r in st:
        n = r["Kernel_Name"].replace("holo::(anonymous namespace)::", "").replace("void ", "")[:70]
        tot[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        cnt[n] += 1
n = len(steps)
print(f"{n} steps: wall {sum(walls) / n:.1f} us, kernel time {sum(sums) / n:.1f} us, idle between kernels {sum(gaps) / n:.1f} us "
      f"over {sum(ngap) / n:.0f} launches ({sum(gaps) / max(sum(ngap), 1):.2f} us per launch)")
for k, v in tot.most_common(16):
    print(f"  {k:70s} {cnt[k] / n:6.1f} launches/step {v / n:8.1f} us/step {v / cnt[k]:7.1f} us each")
