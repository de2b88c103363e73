#!/bin/bash
# usage (GPU box): tools/gpu_pmc_mem.sh <tag> <kernel-name substring> ALGO D H W C B N_ITER REPS "opts_a" "opts_b" ...
# memory-side counters (separate --pmc passes) around tools/probe/ab_probe.py, one process per option string; prints the
# median per dispatch of every kernel whose name contains the substring: 2 x FETCH_SIZE + WRITE_SIZE in GB, the request counters
tag=$1; pat=$2; shift 2
fixed=("${@:1:8}"); shift 8
out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
P="python $PWD/tools/probe/ab_probe.py"
cd /tmp
i=0
for o in "$@"; do
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    d=$out/v${i}_$(echo $set | tr ' ' '_' | cut -c1-30)
    rocprofv3 -f csv --pmc $set -d $d -o c -- $P "${fixed[@]}" "$o" > $d.log 2>&1
  done
  i=$((i+1))
done
cd - > /dev/null
python - "$out" "$pat" "$@" <<'PY'
import csv, glob, statistics, sys, collections
out, pat, variants = sys.argv[1], sys.argv[2], sys.argv[3:]
for i, o in enumerate(variants):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/v%d_*/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                d[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in d.items():
        med = {n: statistics.median(v) for n, v in c.items()}
        gb = (2 * med.get("FETCH_SIZE", 0) + med.get("WRITE_SIZE", 0)) * 1024 / 1e9
        print("[%s] %s n=%d  HBM %.4f GB (read %.4f, written %.4f)  %s" % (o, k, len(next(iter(c.values()))), gb, 2 * med.get("FETCH_SIZE", 0) * 1024 / 1e9, med.get("WRITE_SIZE", 0) * 1024 / 1e9,
              {n: "%.4g" % v for n, v in sorted(med.items()) if n not in ("FETCH_SIZE", "WRITE_SIZE")}))
PY
