#!/bin/bash
# Kernel-time table of one workload (rocprofv3 --kernel-trace --stats), run on the GPU box.
# usage: scripts/quick_prof.sh <tag> [engine] [n] [iterations] [calls] [n_src]
cd "$GRAFT_REPO_ROOT"
tag=$1; shift
O="$GRAFT_REPO_ROOT/gpurun_out/$tag"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" "$@" > "$O/stats.log" 2>&1 || { tail -5 "$O/stats.log"; exit 1; }
cd "$GRAFT_REPO_ROOT"
python scripts/prof_summary.py "$O/stats" > "$O/summary.txt" 2>&1
cat "$O/summary.txt"
