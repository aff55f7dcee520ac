#!/usr/bin/env python3
"""What sits between consecutive MFMAs of a kernel in a gfx950 assembly listing (one line per 16 MFMAs):
M mfma, p v_pk_*, t v_exp/v_rcp, v other VALU, s SALU, r/w ds read/write, G/S global load/store, L scratch load, W waitcnt,
n s_nop, B barrier.   usage: python scripts/mfma_slots.py <file.s> <mangled-name-substring>"""
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r"^(\S*" + re.escape(sys.argv[2]) + r"\S*):", s, re.M)
body = s[m.start():s.index(".end_amdhsa_kernel", m.start())].split("\n")


def cls(t):
    op = t[0]
    if op.startswith("v_mfma"):
        return "M"
    if op.startswith("ds_read"):
        return "r"
    if op.startswith("ds_write"):
        return "w"
    if op.startswith(("global_load", "buffer_load")):
        return "G"
    if op.startswith("scratch_load"):
        return "L"
    if op.startswith(("global_store", "scratch_store")):
        return "S"
    if op.startswith("s_waitcnt"):
        return "W[" + t[1].replace("vmcnt", "vm").replace("lgkmcnt", "lg") + "]"
    if op.startswith("s_nop"):
        return "n"
    if op.startswith("s_barrier"):
        return "B"
    if op.startswith("v_pk"):
        return "p"
    if op.startswith(("v_exp", "v_rcp")):
        return "t"
    if op.startswith("v_"):
        return "v"
    if op.startswith("s_"):
        return "s"
    return "?"


seq = []
for l in body:
    t = l.strip().split()
    if not t or t[0].startswith(";") or t[0].startswith(".") or t[0].endswith(":"):
        continue
    seq.append(cls(t))
first = seq.index("M")
out, cur = [], []
for c in seq[first + 1:]:
    if c == "M":
        out.append(cur)
        cur = []
    else:
        cur.append(c)
out.append(cur[:120])
for g in range(0, (len(out) + 15) // 16):
    print(g, " | ".join("".join(x) if x else "-" for x in out[g * 16:(g + 1) * 16]))
