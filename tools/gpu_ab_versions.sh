#!/bin/bash
# usage (GPU box): tools/gpu_ab_versions.sh OTHER_TREE ALGO D H W C B N_ITER REPS ["opts"]
# Same-box A/B of two SOURCE TREES (this one against a built checkout of another commit under OTHER_TREE, e.g. a
# `git worktree` of last round's HEAD) on one workload: the probe of each tree is run three times, alternating.
other=$1; shift
for rnd in 1 2 3; do
  for tree in "$other" "."; do
    printf "%-10s " "$tree"
    GRAFT_REPO_ROOT=$(realpath $tree) python $tree/tools/probe/ab_probe.py "$@" 2>&1 | grep -v "amdgpu.ids" | grep "best"
  done
done
