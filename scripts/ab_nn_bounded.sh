#!/bin/bash
# the ICP loop on C3 (100k -> 100k, 30 iterations): every pass with coarse minima + certificate (ICPMI_NN_BOUNDED=0)
# against passes searched behind the previous matches (nn_bounded.h); kernel times from rocprofv3, same box, A B A B
cd "$GRAFT_REPO_ROOT"
for v in 0 1 0 1; do
    rm -rf "gpurun_out/nnb_$v"
    (cd /tmp && TMPDIR=/tmp ICPMI_NN_BOUNDED=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/nnb_$v" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" 0 ${1:-100000} 30 3 ${2:-} > "$GRAFT_REPO_ROOT/gpurun_out/nnb_$v.log" 2>&1)
    echo "=== ICPMI_NN_BOUNDED=$v"
    python scripts/prof_summary.py "gpurun_out/nnb_$v" | grep "k_nn_coarse\|k_nn_resolve\|k_nn_bounds\|k_finish\|k_transform\|k_step" | grep -v "coarse<1\|coarse_rows"
done
