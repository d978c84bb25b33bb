#!/bin/bash
# usage: scripts/sweep_build.sh "<EXTRA flags A>" "<EXTRA flags B>" ... -- rebuilds the library with each
# flag set on the GPU box and prints the per-kernel averages of one C3 run (engine $ENGINE, default 0).
set -e
cd "$GRAFT_REPO_ROOT"
# (the product build comes back whatever ends this script; the Makefile's .build_flags stamp covers a SIGKILL: build.py rebuilds)
trap 'rm -f lidar_slam_from_scratch_amd/csrc/capi.o; make -s -C lidar_slam_from_scratch_amd/csrc EXTRA= > /dev/null 2>&1' EXIT
trap 'exit 143' TERM INT HUP
for flags in "$@"; do
    rm -f lidar_slam_from_scratch_amd/csrc/capi.o
    make -s -C lidar_slam_from_scratch_amd/csrc EXTRA="$flags" > /dev/null 2>&1
    tag=$(echo "$flags" | tr -c 'A-Za-z0-9=\n' '_')
    (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/sweep_$tag" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" ${ENGINE:-0} > "$GRAFT_REPO_ROOT/gpurun_out/sweep_$tag.log" 2>&1)
    echo "=== $flags"
    python scripts/prof_summary.py "gpurun_out/sweep_$tag" | head -8
done
# leave the default build behind
rm -f lidar_slam_from_scratch_amd/csrc/capi.o
make -s -C lidar_slam_from_scratch_amd/csrc > /dev/null 2>&1
