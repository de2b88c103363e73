#!/bin/bash
out=gpurun_out/r03i; mkdir -p $out
{
echo "== C2 mid swizzle"; python tools/probe/ab_probe.py admm 1 3040 4056 3 1 40 2 "" "mid_swz=1"
echo "== C1 mid swizzle"; python tools/probe/ab_probe.py admm 1 270 480 3 1 5 20 "" "mid_swz=1"
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $out/ab.log
export TMPDIR=/tmp; here=$PWD; cd /tmp
for v in "" "mid_swz=1"; do
  LPC_OPTIONS="$v" rocprofv3 -f csv --pmc FETCH_SIZE -d $here/$out/fetch_${v:-default} -o c -- python $here/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --n-iter 40 > /dev/null 2>&1
done
cd $here
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r03i/fetch_*")):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_cols_mid_admm" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v.sort(); print(d, k, "median FETCH_SIZE KiB", v[len(v)//2], "-> reads GB", 2*v[len(v)//2]*1024/1e9)
PY
