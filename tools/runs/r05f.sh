mkdir -p gpurun_out
for m in 0 1 2 4 8 3 10 15; do
  if [ $m = 0 ]; then o=""; else o="module_dir=_ab_x/knock$m"; fi
  echo "== knock mask $m" >> gpurun_out/r05f_knock.log
  tools/gpu_ab.sh fista 1 3040 4056 3 1 40 2 "$o" >> gpurun_out/r05f_knock.log 2>&1
done
grep "knock\|best" gpurun_out/r05f_knock.log
