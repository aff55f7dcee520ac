#!/usr/bin/env python3
"""Summarise `HOLO_BENCH_OPS=1 python bench.py ... 2> ops.err` per-op timing dumps side by side."""
import json, sys, collections
def load(fn):
    agg = collections.OrderedDict(); tot = 0.0
    for l in open(fn):
        if not l.startswith('# op'): continue
        o = json.loads(l[5:]); tot += o['ms']
        if o['op'] == 'conv':
            key = f"conv{o['ksz']} {o['kernel'][5:11]} cin{o['cin']} cout{o['cout']} od{o['out_dim']} s{o['stride']}u{int(o['upsample'])} sk{int(o['fused_skip'])}"
        else:
            key = f"{o['op']} {o['cin']} {o['cout']} {o['out_dim']}"
        a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += o['ms']; a[2] += o['flops']
    return agg, tot
tabs = [load(f) for f in sys.argv[1:]]
keys = list(tabs[0][0].keys())
for t in tabs[1:]:
    for k in t[0]:
        if k not in keys: keys.append(k)
for k in keys:
    row = f"{k:58s}"
    for agg, _ in tabs:
        a = agg.get(k)
        row += f" | n{a[0]:2d} {a[1]/a[0]*1e3:7.1f}us {a[2]/max(a[1],1e-9)/1e9:6.1f}TF" if a else " | " + " " * 24
    print(row)
print("sum ms:", [round(t[1], 3) for t in tabs])
