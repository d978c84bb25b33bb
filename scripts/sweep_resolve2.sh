#!/bin/bash
cd "$GRAFT_REPO_ROOT"
# (the product build comes back whatever ends this script; the Makefile's .build_flags stamp covers a SIGKILL: build.py rebuilds)
trap 'rm -f lidar_slam_from_scratch_amd/csrc/capi.o; make -s -C lidar_slam_from_scratch_amd/csrc EXTRA= > /dev/null 2>&1' EXIT
trap 'exit 143' TERM INT HUP
for w in "$@"; do
    rm -f lidar_slam_from_scratch_amd/csrc/capi.o
    make -s -C lidar_slam_from_scratch_amd/csrc EXTRA="-DICPMI_RESOLVE_WAVES=$w" > /dev/null 2>&1
    echo "##### ICPMI_RESOLVE_WAVES=$w"
    for nsrc in 25000 50000; do
        rm -rf "gpurun_out/rs_x"
        (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/rs_x" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" 0 100000 30 3 $nsrc > "$GRAFT_REPO_ROOT/gpurun_out/rs_x.log" 2>&1)
        echo "=== n_src $nsrc"
        python scripts/prof_summary.py "gpurun_out/rs_x" | grep "k_nn_resolve\|k_finish_step"
    done
done
rm -f lidar_slam_from_scratch_amd/csrc/capi.o
make -s -C lidar_slam_from_scratch_amd/csrc > /dev/null 2>&1
