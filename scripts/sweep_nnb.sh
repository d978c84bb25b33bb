#!/bin/bash
# where k_nn_resolve_bounded's time goes: rebuild with -DICPMI_NNB_STOP=<phase> (results are WRONG in those builds:
# timing only; only the FIRST bounded pass sees sane inputs, so 2 iterations) and time the kernel on C3.
cd "$GRAFT_REPO_ROOT"
# (the product build comes back whatever ends this script; the Makefile's .build_flags stamp covers a SIGKILL: build.py rebuilds)
trap 'rm -f lidar_slam_from_scratch_amd/csrc/capi.o; make -s -C lidar_slam_from_scratch_amd/csrc EXTRA= > /dev/null 2>&1' EXIT
trap 'exit 143' TERM INT HUP
for flags in "$@"; do
    rm -f lidar_slam_from_scratch_amd/csrc/capi.o
    make -s -C lidar_slam_from_scratch_amd/csrc EXTRA="$flags" > /dev/null 2>&1
    tag=$(echo "$flags" | tr -c 'A-Za-z0-9=\n' '_')
    rm -rf "gpurun_out/nnbs_$tag"
    (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/nnbs_$tag" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" 0 100000 1 6 > "$GRAFT_REPO_ROOT/gpurun_out/nnbs_$tag.log" 2>&1)
    echo "=== $flags"
    python scripts/prof_summary.py "gpurun_out/nnbs_$tag" | grep "k_nn_resolve_bounded\|k_nn_resolve<16>"
done
rm -f lidar_slam_from_scratch_amd/csrc/capi.o
make -s -C lidar_slam_from_scratch_amd/csrc > /dev/null 2>&1
