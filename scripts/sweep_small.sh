#!/bin/bash
# A/B builds on the two regimes at once: C3 (100k x 100k, k_nn_resolve) and a 12.5k-row shard of it
# (k_nn_resolve4).  usage: scripts/sweep_small.sh "<EXTRA flags>" ...
cd "$GRAFT_REPO_ROOT"
# (the product build comes back whatever ends this script; the Makefile's .build_flags stamp covers a SIGKILL: build.py rebuilds)
trap 'rm -f lidar_slam_from_scratch_amd/csrc/capi.o; make -s -C lidar_slam_from_scratch_amd/csrc EXTRA= > /dev/null 2>&1' EXIT
trap 'exit 143' TERM INT HUP
for flags in "$@"; do
    rm -f lidar_slam_from_scratch_amd/csrc/capi.o
    make -s -C lidar_slam_from_scratch_amd/csrc EXTRA="$flags" > /dev/null 2>&1
    echo "=== $flags"
    for a in "0 100000 30 3" "0 100000 30 3 12500"; do
        rm -rf gpurun_out/ss
        (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/ss" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" $a > "$GRAFT_REPO_ROOT/gpurun_out/ss.log" 2>&1)
        python scripts/prof_summary.py gpurun_out/ss | grep "k_nn_resolve\|k_nn_coarse<0\|k_finish_step" | sed "s/^/[$a] /" | cut -c1-150
    done
done
rm -f lidar_slam_from_scratch_amd/csrc/capi.o
make -s -C lidar_slam_from_scratch_amd/csrc > /dev/null 2>&1
