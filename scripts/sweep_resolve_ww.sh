cd "$GRAFT_REPO_ROOT"
# (the product build comes back whatever ends this script; the Makefile's .build_flags stamp covers a SIGKILL: build.py rebuilds)
trap 'rm -f lidar_slam_from_scratch_amd/csrc/capi.o; make -s -C lidar_slam_from_scratch_amd/csrc EXTRA= > /dev/null 2>&1' EXIT
trap 'exit 143' TERM INT HUP
for ww in 4 8; do
    rm -f lidar_slam_from_scratch_amd/csrc/capi.o
    make -s -C lidar_slam_from_scratch_amd/csrc EXTRA="-DICPMI_RESOLVE_WW=$ww" > /dev/null 2>&1
    for rep in 1 2; do
    rm -rf gpurun_out/ww_$ww
    (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/ww_$ww" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" 0 100000 30 3 > /dev/null 2>&1)
    echo "=== ICPMI_RESOLVE_WW=$ww"
    python scripts/prof_summary.py gpurun_out/ww_$ww | grep "k_nn_resolve_bounded\|k_finish_step_transform"
    rm -rf gpurun_out/ww_$ww
    done
done
rm -f lidar_slam_from_scratch_amd/csrc/capi.o
make -s -C lidar_slam_from_scratch_amd/csrc > /dev/null 2>&1
