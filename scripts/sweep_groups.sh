#!/bin/bash
# Kernel times of the culled engine (C3, 20 iterations, 3 calls) under the coarse kernel's launch knobs (run on the GPU box).
# usage: scripts/sweep_groups.sh <tag>     -> gpurun_out/<tag>/*.txt
cd "$GRAFT_REPO_ROOT"
tag=${1:-sweep_groups}
for cfg in "8 0" "4 0" "4 3" "8 3"; do
  set -- $cfg
  export ICPMI_GROUPS_WAVES=$1 ICPMI_GROUPS_GRID=$2
  bash scripts/quick_prof.sh "$tag/w$1_g$2" 0 100000 20 3 > /dev/null || exit 1
  echo "== waves $1 grid/CU $2 (0: chip's resident set)"
  grep "k_nn_coarse_groups<false\|k_finish_step_transform_cull\|k_nn_resolve_bounded" "gpurun_out/$tag/w$1_g$2/summary.txt"
done
