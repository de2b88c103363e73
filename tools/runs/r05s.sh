#!/bin/bash
# small frames: stamped timeline of the last iteration + kernel-trace gaps
out=gpurun_out/r05s; mkdir -p $out
export TMPDIR=/tmp
for shp in "270 480" "380 507"; do
  t=${shp/ /x}
  python tools/stamp_timeline.py 1 $shp 3 1 5 > $out/stamps_$t.log 2>&1
  python tools/stamp_timeline.py 1 $shp 3 1 100 > $out/stamps_${t}_100it.log 2>&1
  (cd /tmp && rocprofv3 --kernel-trace -f csv -d $GRAFT_REPO_ROOT/$out/trace_$t -o t -- python $GRAFT_REPO_ROOT/tools/probe/ab_probe.py admm 1 $shp 3 1 100 20 "" > $GRAFT_REPO_ROOT/$out/trace_$t.log 2>&1)
  f=$(find $out/trace_$t -name "*kernel_trace.csv" | head -1)
  python tools/trace_gaps.py $f 4000 > $out/gaps_$t.md 2>&1
  rm -rf $out/trace_$t
done
tail -n 30 $out/stamps_270x480.log $out/gaps_270x480.md
