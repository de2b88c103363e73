mkdir -p gpurun_out
for m in "" r0 u0 w5 r0w5 ""; do
  if [ -z "$m" ]; then o=""; else o="module_dir=_ab_x/knock$m"; fi
  echo "== variant '$m'" >> gpurun_out/r05o_tc.log
  tools/gpu_ab.sh fista 1 3040 4056 3 1 40 2 "$o" >> gpurun_out/r05o_tc.log 2>&1
done
grep "variant\|best" gpurun_out/r05o_tc.log
