mkdir -p gpurun_out
for m in 0 16 32 48 0 48; do
  if [ $m = 0 ]; then o=""; else o="module_dir=_ab_x/knock$m"; fi
  echo "== knock mask $m" >> gpurun_out/r05m_knock_tw.log
  tools/gpu_ab.sh fista 1 3040 4056 3 1 40 2 "$o" >> gpurun_out/r05m_knock_tw.log 2>&1
done
grep "knock\|best" gpurun_out/r05m_knock_tw.log
