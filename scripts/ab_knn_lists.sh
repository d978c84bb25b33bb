#!/bin/bash
# normal estimation on the 100k cloud (and a 9k LiDAR-like frame): round 2's slot-minimum form
# (ICPMI_KNN_LISTS=0) against the list form (knn_lists.h), kernel times from rocprofv3, same box
cd "$GRAFT_REPO_ROOT"
for v in 0 1 0 1; do
    rm -rf "gpurun_out/knnl_$v"
    (cd /tmp && TMPDIR=/tmp ICPMI_KNN_LISTS=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/knnl_$v" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" 0 100000 2 3 > "$GRAFT_REPO_ROOT/gpurun_out/knnl_$v.log" 2>&1)
    echo "=== ICPMI_KNN_LISTS=$v"
    python scripts/prof_summary.py "gpurun_out/knnl_$v" | grep "k_knn\|k_nn_coarse<1\|k_nn_coarse_rows\|k_normals\|k_gather_rows\|k_scatter"
done
