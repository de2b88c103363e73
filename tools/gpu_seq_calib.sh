#!/bin/bash
# usage (GPU box): tools/gpu_seq_calib.sh <tag> -- memory-side counters of tools/probe/seq_pattern (the access pattern of
# C4's sequential middle on a known byte count) in separate --pmc passes; summary on stdout and in gpurun_out/<tag>/calib.txt
tag=${1:-seqcal}
out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
BIN=$PWD/tools/probe/seq_pattern
[ -x $BIN ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $PWD/tools/probe/seq_pattern.hip -o $BIN
cd /tmp
for mode in 0 1 2 3; do
  $BIN $mode 64 20 > $out/run_$mode.txt 2>&1
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"; do
    d=$out/m${mode}_$(echo $set | tr ' ' '_' | cut -c1-40)
    rocprofv3 -f csv --pmc $set -d $d -o c -- $BIN $mode 64 6 > $d.log 2>&1
  done
done
cd - > /dev/null
python - "$out" <<'PY' | tee $out/calib.txt
import csv, glob, statistics, sys, collections, os
out = sys.argv[1]
for mode in range(4):
    print(open(os.path.join(out, "run_%d.txt" % mode)).read().strip())
    c = collections.defaultdict(list)
    for f in glob.glob(out + "/m%d_*/**/*counter_collection.csv" % mode, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_tiles" in r["Kernel_Name"] or "k_stream" in r["Kernel_Name"]:
                c[r["Counter_Name"]].append(float(r["Counter_Value"]))
    med = {k: statistics.median(v) for k, v in c.items()}
    print("   ", {k: "%.4g" % v for k, v in sorted(med.items())})
    if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
        print("    FETCH_SIZE KiB -> %.4f GB (x2: %.4f GB)   WRITE_SIZE KiB -> %.4f GB" % (med["FETCH_SIZE"] * 1024 / 1e9, 2 * med["FETCH_SIZE"] * 1024 / 1e9, med["WRITE_SIZE"] * 1024 / 1e9))
PY
