#!/bin/bash
# Counter passes of one workload with a SHORT timeout per pass (a group the hardware cannot collect makes rocprofv3 abort
# and hang until the timeout).  usage: scripts/pmc_passes.sh <tag> "<run_align_once args>" "<group>" ["<group>" ...]
cd "$GRAFT_REPO_ROOT"
tag=$1; args=$2; shift 2
O="$GRAFT_REPO_ROOT/gpurun_out/$tag"
mkdir -p "$O"
i=0
for grp in "$@"; do
    i=$((i+1))
    (cd /tmp && TMPDIR=/tmp timeout -k 5 90 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$O/pmc_$i" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" $args > "$O/pmc_$i.log" 2>&1) || { echo "pass $i ($grp) failed"; grep -m1 "error code" "$O/pmc_$i.log"; }
done
python scripts/prof_summary.py "$O" > "$O/summary.txt" 2>&1
grep -v "rocprim\|__amd_rocclr\|k_bbox\|k_morton\|k_gather\|k_split_frames\|k_pack" "$O/summary.txt"
