#!/usr/bin/env python3
"""gpurun_out/<tag>/pmc_sq.csv + pmc_grbm.csv (scripts/gpu_pmc.sh / gpu_pmc_side.sh) -> profiles/<name>_pmc_summary.json:
per (kernel, grid) the averaged counters of one launch and the derived figures DESIGN.md quotes.

  mfma_pipe_busy          SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x launch cycles), launch cycles = GRBM_GUI_ACTIVE / 8 XCDs,
                          SIMDs = 256 CUs x 4  (1.0 = every matrix pipe busy for the whole launch)
  wave_time_in_waitcnt    SQ_WAIT_ANY / SQ_WAVE_CYCLES            (waves parked at s_waitcnt / barriers)
  wave_time_issue_stalled SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES       (waves that could not issue: pipe busy, MFMA dependencies)
  valu_active             4 x SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES (the counters tick in quad-cycles)
  lds_bank_conflict       SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE when collected, else / (launch cycles x 256 CUs)
Separate --pmc passes, as MI355X_MICROARCH.md prescribes; a kernel / grid missing from one pass keeps what the other has.
usage: python scripts/pmc_summary.py <gpurun_out tag> <output json> [kernel-name substring ...]"""
import csv
import json
import os
import sys

tag, out = sys.argv[1], sys.argv[2]
filters = sys.argv[3:] or ["conv_", "flash_attn", "render", "gemm_kernel", "gn_finalize", "splitk_reduce", "wgrad"]
agg = {}
for name in ("sq", "grbm"):
    fn = f"gpurun_out/{tag}/pmc_{name}.csv"
    if not os.path.isfile(fn):
        continue
    for r in csv.DictReader(open(fn)):
        if not any(f in r["kernel"] for f in filters):
            continue
        key = f'{r["kernel"][:100]} grid {r["grid"]}'
        d = agg.setdefault(key, {"dispatches": int(r["dispatches"])})
        d[r["counter"]] = float(r["avg"])
for key, d in agg.items():
    g = d.get("GRBM_GUI_ACTIVE")
    if g and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        d["mfma_pipe_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (g / 8.0 * 1024.0)
    wc = d.get("SQ_WAVE_CYCLES")
    if wc:
        if "SQ_WAIT_ANY" in d:
            d["wave_time_in_waitcnt"] = d["SQ_WAIT_ANY"] / wc
        if "SQ_WAIT_INST_ANY" in d:
            d["wave_time_issue_stalled"] = d["SQ_WAIT_INST_ANY"] / wc
    if d.get("SQ_BUSY_CYCLES") and "SQ_ACTIVE_INST_VALU" in d:
        d["valu_active"] = 4.0 * d["SQ_ACTIVE_INST_VALU"] / d["SQ_BUSY_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in d:
        if d.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict"] = d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]
        elif g:
            d["lds_bank_conflict_cycles_per_cu_over_active"] = d["SQ_LDS_BANK_CONFLICT"] / (g / 8.0 * 256.0)
res = {"source": f"rocprofv3 --pmc passes of gpurun_out/{tag} (scripts/gpu_pmc.sh / gpu_pmc_side.sh), averaged per launch",
       "kernels": dict(sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0) * kv[1]["dispatches"]))}
json.dump(res, open(out, "w"), indent=1)
for k, d in list(res["kernels"].items())[:14]:
    print(f'{k[-78:]:78s} x{d["dispatches"]:4d}  mfma busy {d.get("mfma_pipe_busy", float("nan")):.3f}  waitcnt '
          f'{d.get("wave_time_in_waitcnt", float("nan")):.2f}  issue-stalled {d.get("wave_time_issue_stalled", float("nan")):.2f}')
