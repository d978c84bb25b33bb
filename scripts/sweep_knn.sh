#!/bin/bash
# where k_knn_resolve's time goes: rebuild with -DICPMI_KNN_STOP=<phase> (results are WRONG in those
# builds: timing only) and time the normals kernels on the 100k cloud.
cd "$GRAFT_REPO_ROOT"
# (the product build comes back whatever ends this script; the Makefile's .build_flags stamp covers a SIGKILL: build.py rebuilds)
trap 'rm -f lidar_slam_from_scratch_amd/csrc/capi.o; make -s -C lidar_slam_from_scratch_amd/csrc EXTRA= > /dev/null 2>&1' EXIT
trap 'exit 143' TERM INT HUP
for flags in "$@"; do
    rm -f lidar_slam_from_scratch_amd/csrc/capi.o
    make -s -C lidar_slam_from_scratch_amd/csrc EXTRA="$flags" > /dev/null 2>&1
    tag=$(echo "$flags" | tr -c 'A-Za-z0-9=\n' '_')
    rm -rf "gpurun_out/knn_$tag"
    (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/knn_$tag" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" 0 100000 2 3 > "$GRAFT_REPO_ROOT/gpurun_out/knn_$tag.log" 2>&1)
    echo "=== $flags"
    python scripts/prof_summary.py "gpurun_out/knn_$tag" | grep "k_knn_resolve\|k_nn_coarse<1\|k_normals"
done
rm -f lidar_slam_from_scratch_amd/csrc/capi.o
make -s -C lidar_slam_from_scratch_amd/csrc > /dev/null 2>&1
