#!/bin/bash
# Kernel-level profile of one registration of the C2 stand-in at the size BASELINE.json configs[1] names (~18k x 20k points,
# the LiDAR-like pair at a 0.3 m voxel), on the GPU box: kernel_stats.csv + the last call as a timeline with its per-
# iteration breakdown (scripts/call_timeline.py).   usage: scripts/prof_c2.sh <tag> [engine] [voxel]
cd "$GRAFT_REPO_ROOT"
tag=$1; eng=${2:-0}; vox=${3:-0.3}
O="$GRAFT_REPO_ROOT/gpurun_out/$tag"
mkdir -p "$O"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$GRAFT_REPO_ROOT/scripts/run_c2_once.py" "$eng" 3 "$vox" > "$O/run.log" 2>&1) || { tail -5 "$O/run.log"; exit 1; }
python scripts/prof_summary.py "$O/stats" > "$O/summary.txt" 2>&1
python scripts/call_timeline.py "$O/stats" > "$O/timeline.txt" 2>&1
tail -1 "$O/run.log"; head -14 "$O/summary.txt" | cut -c1-140; tail -22 "$O/timeline.txt"
