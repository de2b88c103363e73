mkdir -p gpurun_out
for H in 380 760 1520 3040; do
  tools/gpu_ab.sh fista 1 $H 4056 3 1 40 2 "" >> gpurun_out/r05i_rows_vs_H.log 2>&1
done
grep "best\|padded" gpurun_out/r05i_rows_vs_H.log | cut -c1-250
