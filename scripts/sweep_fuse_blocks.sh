cd "$GRAFT_REPO_ROOT"
for fb in 32 64 128 256; do
    rm -rf gpurun_out/fb_$fb
    (cd /tmp && TMPDIR=/tmp ICPMI_FUSE_BLOCKS=$fb timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/fb_$fb" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" 0 100000 30 3 > "$GRAFT_REPO_ROOT/gpurun_out/fb_$fb.log" 2>&1)
    echo "=== ICPMI_FUSE_BLOCKS=$fb"
    python scripts/prof_summary.py "gpurun_out/fb_$fb" | grep "k_finish_step_transform\|k_nn_resolve_bounded"
    rm -rf gpurun_out/fb_$fb
done
