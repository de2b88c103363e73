#!/bin/bash
# usage (GPU box): tools/gpu_pmc_ab.sh <tag> <kernel-name substring> ALGO D H W C B N_ITER REPS "opts_a" "opts_b" ...
# rocprofv3 kernel trace + two PMC passes around tools/probe/ab_probe.py; prints, per kernel whose name contains the
# substring, the dispatch count, median duration and median of each counter (raw output under gpurun_out/<tag>/).
tag=$1; pat=$2; shift 2
out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
P="python $PWD/tools/probe/ab_probe.py"
cd /tmp
rocprofv3 --kernel-trace -f csv -d $out/trace -o t -- $P "$@" > $out/trace.log 2>&1
rocprofv3 -f csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES -d $out/pmc1 -o c -- $P "$@" > $out/pmc1.log 2>&1
rocprofv3 -f csv --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_ANY -d $out/pmc2 -o c -- $P "$@" > $out/pmc2.log 2>&1
cd - > /dev/null
python - "$out" "$pat" <<'PY'
import csv, glob, statistics, sys, collections
out, pat = sys.argv[1], sys.argv[2]
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    d = collections.defaultdict(list); meta = {}
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            k = (r["Kernel_Name"][:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))
            d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            meta[k] = {x: r.get(x) for x in ("LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size_X", "Grid_Size_X")}
    for k, v in d.items():
        print("trace", k, "n", len(v), "median us %.1f" % statistics.median(v), meta[k])
for p in ("pmc1", "pmc2"):
    for f in glob.glob(out + "/" + p + "/**/*counter_collection.csv", recursive=True):
        d = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                d[(r["Kernel_Name"][:60], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in d.items():
            print(p, k, {n: "%.3g" % statistics.median(v) for n, v in c.items()})
PY
