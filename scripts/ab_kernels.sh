#!/bin/bash
# Same-box kernel-time comparison of builds of the library (rocprofv3 --kernel-trace --stats of scripts/run_align_once.py),
# alternating.  usage: scripts/ab_kernels.sh <tag> <reps> <lib.so|product> [<lib.so|product> ...] -- <run_align_once args>
cd "$GRAFT_REPO_ROOT"
tag=$1; reps=$2; shift 2
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
shift
O="$GRAFT_REPO_ROOT/gpurun_out/$tag"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for rep in $(seq 1 $reps); do
    for lib in "${libs[@]}"; do
        name=$(basename $lib .so)
        rm -rf "$O/stats_$name"
        if [ "$lib" = product ]; then
            timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$name" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" "$@" > "$O/log_$name.txt" 2>&1 || { tail -5 "$O/log_$name.txt"; exit 1; }
        else
            timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$name" -- python3 "$GRAFT_REPO_ROOT/scripts/with_lib.py" "$GRAFT_REPO_ROOT/$lib" "$GRAFT_REPO_ROOT/scripts/run_align_once.py" "$@" > "$O/log_$name.txt" 2>&1 || { tail -5 "$O/log_$name.txt"; exit 1; }
        fi
        echo "== $name (rep $rep)"
        python3 "$GRAFT_REPO_ROOT/scripts/prof_summary.py" "$O/stats_$name" 2>&1 | sed -n 2,7p | cut -c1-60,88-140
    done
done
