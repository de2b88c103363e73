#!/bin/bash
# usage: tools/gpu_ab.sh <outdir> "<label>:<ENV=VAL ...>" ... -- on the GPU box: short default-bench runs under different
# environment knobs, one summary line each
out=gpurun_out/$1; shift; mkdir -p $out
export MPLBACKEND=Agg
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs $BENCH_ARGS"
for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo -n "== $label [$envs] : "
  env $envs $B > $out/bench_$label.json 2> $out/bench_$label.log && python tools/kernel_summary.py $out/bench_$label.json || tail -3 $out/bench_$label.log
done
