#!/usr/bin/env python3
"""Instruction mix of the basic blocks of one kernel in a gfx950 assembly listing.
usage: python scripts/asm_loops.py <file.s> <mangled-name-substring> [min_instructions]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 60
m = re.search(r"^(\S*" + re.escape(key) + r"\S*):", s, re.M)
i = m.start()
j = s.index(".end_amdhsa_kernel", i)
body = s[i:j].split("\n")


def cls(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_pk_"):
        return "v_pk"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


cur = (0, "entry")
counts = collections.OrderedDict()
counts[cur] = collections.Counter()
for n, l in enumerate(body):
    mm = re.match(r"(\.LBB\d+_\d+):", l)
    if mm:
        cur = (n, mm.group(1))
        counts[cur] = collections.Counter()
        continue
    t = l.strip().split()
    if not t or t[0].startswith(";") or t[0].startswith("."):
        continue
    counts[cur][cls(t[0])] += 1
    if t[0].startswith("s_cbranch") and len(t) > 1 and t[1] == cur[1]:  # back edge: what follows is a new region
        cur = (n, cur[1] + "+after")
        counts[cur] = collections.Counter()
for k, c in counts.items():
    if sum(c.values()) >= minn:
        print(k, dict(c))
