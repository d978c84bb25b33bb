#!/bin/bash
# usage: scripts/sweep_run.sh "<args for run_align_once.py>" ... -- profiles the CURRENT build once per
# argument string and prints the top kernels.
cd "$GRAFT_REPO_ROOT"
i=0
for a in "$@"; do
    i=$((i+1))
    (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/run_$i" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" $a > "$GRAFT_REPO_ROOT/gpurun_out/run_$i.log" 2>&1)
    echo "=== $a"
    python scripts/prof_summary.py "gpurun_out/run_$i" | head -${TOPK:-9}
done
