#!/bin/bash
# A/B of the resolve kernels: rebuild with -DICPMI_RESOLVE_WAVES=<w> and time k_nn_resolve* on
# C3 (all-pairs, pruned), one 12.5k-row shard of it, and the C2 pair.
cd "$GRAFT_REPO_ROOT"
# (the product build comes back whatever ends this script; the Makefile's .build_flags stamp covers a SIGKILL: build.py rebuilds)
trap 'rm -f lidar_slam_from_scratch_amd/csrc/capi.o; make -s -C lidar_slam_from_scratch_amd/csrc EXTRA= > /dev/null 2>&1' EXIT
trap 'exit 143' TERM INT HUP
for w in "$@"; do
    rm -f lidar_slam_from_scratch_amd/csrc/capi.o
    make -s -C lidar_slam_from_scratch_amd/csrc EXTRA="-DICPMI_RESOLVE_WAVES=$w" > /dev/null 2>&1
    echo "##### ICPMI_RESOLVE_WAVES=$w"
    i=0
    for a in "0" "3" "0 100000 30 3 12500"; do
        i=$((i+1))
        rm -rf "gpurun_out/rs_$i"
        (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/rs_$i" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" $a > "$GRAFT_REPO_ROOT/gpurun_out/rs_$i.log" 2>&1)
        echo "=== run_align_once $a"
        python scripts/prof_summary.py "gpurun_out/rs_$i" | grep "k_nn_resolve\|k_finish_step"
    done
    rm -rf gpurun_out/rs_c2
    (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/rs_c2" -- python3 "$GRAFT_REPO_ROOT/scripts/run_c2_once.py" 0 > "$GRAFT_REPO_ROOT/gpurun_out/rs_c2.log" 2>&1)
    echo "=== C2"; grep "ms per call" gpurun_out/rs_c2.log
    python scripts/prof_summary.py gpurun_out/rs_c2 | grep "k_nn_resolve\|k_finish_step"
done
# leave the default build behind
rm -f lidar_slam_from_scratch_amd/csrc/capi.o
make -s -C lidar_slam_from_scratch_amd/csrc > /dev/null 2>&1
