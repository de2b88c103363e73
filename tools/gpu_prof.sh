#!/bin/bash
# usage: tools/gpu_prof.sh <tag> [extra bench args] -- on the GPU box: rocprofv3 kernel stats + separate PMC passes of a
# short bench run (ONE 40-iteration call after a 40-iteration warm-up: 36 of every 40 dispatches of a hot-loop kernel are
# steady-state); raw output under gpurun_out/<tag>/prof_*, condensed by tools/summarize_prof.py <tag> gpurun_out/<tag>
tag=$1; shift
out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
B="python $PWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --n-iter 40 $@"
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $out/prof_stats -o stats -- $B > $out/prof_bench.json 2> $out/prof_stats.log
rocprofv3 -f csv --pmc FETCH_SIZE -d $out/prof_fetch -o c -- $B > $out/prof_fetch.log 2>&1
rocprofv3 -f csv --pmc WRITE_SIZE -d $out/prof_write -o c -- $B > $out/prof_write.log 2>&1
rocprofv3 -f csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $out/prof_sq -o c -- $B > $out/prof_sq.log 2>&1
rocprofv3 -f csv --pmc TCC_HIT_sum TCC_MISS_sum -d $out/prof_l2 -o c -- $B > $out/prof_l2.log 2>&1
cd - > /dev/null
ls $out
