#!/bin/bash
# k_nn_resolve_bounded with the register allocation forced to admit 5 / 6 / 7 waves per SIMD (-DICPMI_BOUNDED_OCC): C3, rocprofv3
cd "$GRAFT_REPO_ROOT"
# (the product build comes back whatever ends this script; the Makefile's .build_flags stamp covers a SIGKILL: build.py rebuilds)
trap 'rm -f lidar_slam_from_scratch_amd/csrc/capi.o; make -s -C lidar_slam_from_scratch_amd/csrc EXTRA= > /dev/null 2>&1' EXIT
trap 'exit 143' TERM INT HUP
for occ in 1 6 7 8; do
    rm -f lidar_slam_from_scratch_amd/csrc/capi.o
    make -s -C lidar_slam_from_scratch_amd/csrc EXTRA="-DICPMI_BOUNDED_OCC=$occ" > /dev/null 2>&1
    rm -rf gpurun_out/occ_$occ
    (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/occ_$occ" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" 0 100000 30 3 > /dev/null 2>&1)
    echo "=== ICPMI_BOUNDED_OCC=$occ"
    python scripts/prof_summary.py gpurun_out/occ_$occ | grep "k_nn_resolve_bounded"
    rm -rf gpurun_out/occ_$occ
done
rm -f lidar_slam_from_scratch_amd/csrc/capi.o
make -s -C lidar_slam_from_scratch_amd/csrc > /dev/null 2>&1
