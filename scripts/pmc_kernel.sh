#!/bin/bash
# PMC counters of one workload, one counter group per pass (rocprofv3 --pmc with --kernel-trace only).
# usage: scripts/pmc_kernel.sh <tag> "<run_align_once args>" "<counter group>" ["<counter group>" ...]
cd "$GRAFT_REPO_ROOT"
tag=$1; args=$2; shift 2
O="$GRAFT_REPO_ROOT/gpurun_out/$tag"
mkdir -p "$O"
i=0
for grp in "$@"; do
    i=$((i+1))
    (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$O/pmc_$i" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" $args > "$O/pmc_$i.log" 2>&1) || { tail -5 "$O/pmc_$i.log"; exit 1; }
done
python scripts/prof_summary.py "$O" > "$O/summary.txt" 2>&1
grep -v "rocprim\|__amd_rocclr\|k_bbox\|k_morton\|k_gather\|k_split_frames\|k_pack" "$O/summary.txt"
