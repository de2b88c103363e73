#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/prof_*) into the tracked profiles/ directory.

usage: tools/summarize_prof.py <round-tag> [gpurun_out]
Writes profiles/<tag>_kernel_stats.csv (verbatim --stats summary), profiles/<tag>_counters.md and merges the HBM
traffic per launch of every hot-loop kernel into profiles/traffic.json, keyed by the plan module the profiled run used
(HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE is doubled as MI355X_MICROARCH.md section HBM prescribes for
gfx950).  Per-dispatch values are condensed with the MEDIAN: the profiled command is a 40-iteration call after a
40-iteration warm-up, so 72 of the 80 dispatches of a hot-loop kernel are steady-state ones (the first iteration of a
call skips V_old, the last three run without the sensor-window skips) and the median is their value.
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
         "Separate `--pmc` passes (FETCH_SIZE | WRITE_SIZE | SQ_* | TCC_*), `bench.py --n-iter 40 --steps 1 --warmup 1`, MEDIAN per "
         "dispatch (= the steady-state iterations of a call: 72 of 80 dispatches).",
         "FETCH_SIZE / WRITE_SIZE are KiB; `HBM GB` = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction, guide section HBM).", ""]
cols = ["FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
        "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "TCC_HIT_sum", "TCC_MISS_sum"]
lines.append("| kernel | wg | LDS B | VGPR | SGPR | n | HBM GB | " + " | ".join(cols) + " |")
lines.append("|---|---|---|---|---|---|---|" + "---|" * len(cols))
traffic = {}
ndisp = {}
for k in sorted(agg):
    if not k.startswith("k_"):
        continue
    d = agg[k]
    mean = {c: (sorted(v)[len(v) // 2] if v else None) for c, v in d.items()}      # median, see the docstring
    n = max(len(v) for v in d.values())
    hbm = None
    if mean.get("FETCH_SIZE") is not None and mean.get("WRITE_SIZE") is not None:
        hbm = (2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024
        traffic[k] = hbm
        ndisp[k] = n
    wg, lds, vg, sg = meta[k]
    lines.append(f"| {k} | {wg} | {lds} | {vg} | {sg} | {n} | {hbm / 1e9:.3f} | " if hbm else
                 f"| {k} | {wg} | {lds} | {vg} | {sg} | {n} | - | ")
    lines[-1] += " | ".join(f"{mean[c]:.3g}" if mean.get(c) is not None else "-" for c in cols) + " |"
open(os.path.join(out, f"{tag}_counters.md"), "w").write("\n".join(lines) + "\n")


def kid_of(k):
    """hot-loop kernel name -> bench.py's kernel id (lpc_kernel_id); None for set-up / layout kernels"""
    if k.startswith("k_admm_spatial"):
        return "spatial"
    if k.startswith(("k_admm_rows_fused", "k_rfwd_arrays_x")) or (k.startswith(("k_rfwd_half", "k_rfwd_arrays")) and "SPlan" in k):
        return "row_fwd"
    if k.startswith(("k_rinv_half", "k_rinv_arrays")) and "SPlan" in k:
        return "row_inv"
    if k.startswith(("k_cols_mid_admm", "k_cols_mid_mul")):
        return "col_mid"
    if k.startswith(("k_rinv_gd_update", "k_gd_update_fwd_v2")):   # gradient-descent family: inverse rows + update (+ next forward rows)
        return "spatial"
    if k.startswith(("k_rinv_gd_mid", "k_gd_resid_v2")):          # ... inverse rows + residual + forward rows
        return "row_inv"
    if k.startswith("k_cols<") and "SPlan" in k:
        return "col_a_inv" if k.split(",")[2].strip() == "true" else "col_a_fwd"
    return None


bench_json = os.path.join(src, "prof_bench.json")
plan = None
if os.path.exists(bench_json):
    for ln in open(bench_json):
        if ln.startswith("{"):
            bj = json.loads(ln)
            plan = bj.get("engine_plan", "")
            workload = bj.get("config", {}).get("workload", "")
if plan and "plan module " in plan:
    key = plan.split("plan module ")[1].strip()
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from lenslesspicam_amd import build as _build

    # bench.py reports an entry only while the kernels are the ones it was measured on (load_traffic)
    entry = {"plan_module": key, "source_fingerprint": _build.fingerprint(), "workload": workload, "snapshot": tag, "n_iter": 40,
             "source": f"profiles/{tag}_counters.md",
             "formula": "(2*FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes, median per dispatch", "kernels": {}}
    for k, v in sorted(traffic.items(), key=lambda kv: ndisp.get(kv[0], 0)):      # the hot-loop kernel of an id is the
        kid = kid_of(k)                                                              # one dispatched most often (set-up
        if kid:                                                                      # transforms share some ids)
            entry["kernels"][kid] = {"kernel": k, "hbm_bytes_per_launch": v}
    tf = os.path.join(out, "traffic.json")
    tj = json.load(open(tf)) if os.path.exists(tf) else {"plans": []}
    # entries of other source trees are stale (bench.py ignores them): they leave the file
    tj["plans"] = [e for e in tj["plans"] if e.get("plan_module") != key and e.get("source_fingerprint") == entry["source_fingerprint"]] + [entry]
    json.dump(tj, open(tf, "w"), indent=1)
print("\n".join(lines))
