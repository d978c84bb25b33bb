#!/bin/bash
# Whole-split scans of the resolve's certificate: loads of 1 / 2 / 4 / 8 rounds in flight (-DICPMI_RESOLVE4_SCANBATCH,
# -DICPMI_RESOLVE_SCANBATCH).  Per build: k_nn_resolve4<8> on 40 LiDAR-like frames, k_nn_resolve<16> on C3.
# usage (GPU box): bash scripts/sweep_scan.sh "<flags A>" "<flags B>" ...
cd "$GRAFT_REPO_ROOT"
# (the product build comes back whatever ends this script; the Makefile's .build_flags stamp covers a SIGKILL: build.py rebuilds)
trap 'rm -f lidar_slam_from_scratch_amd/csrc/capi.o; make -s -C lidar_slam_from_scratch_amd/csrc EXTRA= > /dev/null 2>&1' EXIT
trap 'exit 143' TERM INT HUP
python - <<PY
import sys; sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import run_sequence
run_sequence.write_synthetic_drive("/tmp/drive_small", 0, 40)
PY
for flags in "$@"; do
    rm -f lidar_slam_from_scratch_amd/csrc/capi.o
    make -s -C lidar_slam_from_scratch_amd/csrc EXTRA="$flags" > /dev/null 2>&1 || exit 1
    tag=$(echo "$flags" | tr -c 'A-Za-z0-9=\n' '_')
    O="$GRAFT_REPO_ROOT/gpurun_out/scan_$tag"
    mkdir -p "$O"
    (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/seq" -- python3 "$GRAFT_REPO_ROOT/scripts/run_sequence.py" --data_dir /tmp/drive_small > "$O/seq.json" 2> /dev/null) || exit 1
    (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/c3" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" 2 > "$O/c3.log" 2>&1) || exit 1
    echo "=== $flags"
    python scripts/prof_summary.py "$O/seq" | grep "k_nn_resolve4\|k_nn_coarse<0\|k_knn_resolve\|k_nn_coarse<1\|k_knn_exact\|k_normals"
    python scripts/prof_summary.py "$O/c3" | grep "k_nn_resolve<"
    python -c "import json; d=json.load(open('$O/seq.json')); print('   ms per frame %.3f median %.3f' % (d['gpu_ms_per_frame_file_to_pose'], d['gpu_frame_ms_median']))"
done
rm -f lidar_slam_from_scratch_amd/csrc/capi.o
make -s -C lidar_slam_from_scratch_amd/csrc > /dev/null 2>&1
