#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/prof_*) into the tracked profiles/ directory.

usage: tools/summarize_prof.py <round-tag> [gpurun_out]
Writes profiles/<tag>_kernel_stats.csv (verbatim --stats summary), profiles/<tag>_counters.md and,
for the ADMM prox/update kernel, profiles/k1_traffic.json (HBM bytes per launch from the PMC
passes: FETCH_SIZE is doubled as MI355X_MICROARCH.md section HBM prescribes for gfx950, units KiB).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)


def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0]


stats = glob.glob(os.path.join(src, "prof_stats", "*", "*_kernel_stats.csv")) + \
    glob.glob(os.path.join(src, "prof_stats", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(out, f"{tag}_kernel_stats.csv"))

agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for d in sorted(glob.glob(os.path.join(src, "prof_*"))):
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")) + glob.glob(os.path.join(d, "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])

lines = [f"# rocprofv3 PMC summary ({tag})", "",
         "Separate `--pmc` passes (FETCH_SIZE | WRITE_SIZE | SQ_*), `bench.py --n-iter 4`, mean per dispatch.",
         "FETCH_SIZE / WRITE_SIZE are KiB; `HBM GB` = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction, guide section HBM).", ""]
cols = ["FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
        "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "TCC_HIT_sum", "TCC_MISS_sum"]
lines.append("| kernel | wg | LDS B | VGPR | SGPR | n | HBM GB | " + " | ".join(cols) + " |")
lines.append("|---|---|---|---|---|---|---|" + "---|" * len(cols))
traffic = {}
for k in sorted(agg):
    if not k.startswith("k_"):
        continue
    d = agg[k]
    mean = {c: (sum(v) / len(v) if v else None) for c, v in d.items()}
    n = max(len(v) for v in d.values())
    hbm = None
    if mean.get("FETCH_SIZE") is not None and mean.get("WRITE_SIZE") is not None:
        hbm = (2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024
        traffic[k] = hbm
    wg, lds, vg, sg = meta[k]
    lines.append(f"| {k} | {wg} | {lds} | {vg} | {sg} | {n} | {hbm / 1e9:.3f} | " if hbm else
                 f"| {k} | {wg} | {lds} | {vg} | {sg} | {n} | - | ")
    lines[-1] += " | ".join(f"{mean[c]:.3g}" if mean.get(c) is not None else "-" for c in cols) + " |"
open(os.path.join(out, f"{tag}_counters.md"), "w").write("\n".join(lines) + "\n")
for k, v in traffic.items():
    # LPC_K_SPATIAL: the tiled kernel; only when the whole image-domain work is fused into the rows (LPC_FUSE_ROWS) is
    # it k_admm_rows_fused (which otherwise is the forward row kernel carrying the X half)
    if k.startswith("k_admm_spatial") or (k.startswith("k_admm_rows_fused") and not any(
            q.startswith("k_admm_spatial") for q in traffic)):
        json.dump({"kernel": k, "hbm_bytes_per_launch": v, "source": f"profiles/{tag}_counters.md", "snapshot": tag,
                   "formula": "(2*FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes"},
                  open(os.path.join(out, "k1_traffic.json"), "w"), indent=1)
print("\n".join(lines))
