#!/bin/bash
# usage: sweep_flags_engines.sh "<flags>" ... : rebuild per flag set, run engine_compare (100k), print summary
cd "$GRAFT_REPO_ROOT"
# (the product build comes back whatever ends this script; the Makefile's .build_flags stamp covers a SIGKILL: build.py rebuilds)
trap 'rm -f lidar_slam_from_scratch_amd/csrc/capi.o; make -s -C lidar_slam_from_scratch_amd/csrc EXTRA= > /dev/null 2>&1' EXIT
trap 'exit 143' TERM INT HUP
for flags in "$@"; do
    rm -f lidar_slam_from_scratch_amd/csrc/capi.o
    make -s -C lidar_slam_from_scratch_amd/csrc EXTRA="$flags" > /dev/null 2>&1
    echo "##### $flags"
    timeout -k 10 200 python scripts/engine_compare.py 2>&1 | grep "mfma_\|ms_per_call\|it_per_s\|loop_ms\|coarse_ms\|pruned_frac\|hist_max"
done
rm -f lidar_slam_from_scratch_amd/csrc/capi.o
make -s -C lidar_slam_from_scratch_amd/csrc > /dev/null 2>&1
