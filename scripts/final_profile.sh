#!/bin/bash
# Round-end measurement set, run on the GPU box: bench line, kernel stats of the same command,
# PMC passes (one counter group per pass), secondary workloads.  Output under gpurun_out/final/;
# the summaries are then copied into profiles/<round>/ (see profiles/README.md).
# usage: scripts/final_profile.sh a   (bench line, kernel stats, counter passes, secondary workloads)
#        scripts/final_profile.sh b   (frame stream, A/B runs, small-cloud regime, sensitivity)
#        scripts/final_profile.sh c   (round 3: lists / bounded passes A/B, normals timing, fuzz runs)   -- three gpurun calls of <= 20 min
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/final"
mkdir -p "$O"
part=${1:-a}
set -x
if [ "$part" = "a" ]; then
timeout -k 10 400 python bench.py > "$O/bench.json" 2> "$O/bench.err" || exit 1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-pruned-extra > "$O/bench_20.json" 2> "$O/bench_20.err" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-traffic > "$O/stats.log" 2>&1 || exit 1
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
    tag=$(echo "$grp" | cut -d' ' -f1 | tr 'A-Z' 'a-z')
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$O/pmc_$tag" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" 0 100000 30 2 > "$O/pmc_$tag.log" 2>&1 || exit 1
done
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python scripts/bench_extra.py --c4 > "$O/bench_extra.json" 2> "$O/bench_extra.err" || exit 1
timeout -k 10 200 python scripts/engine_compare.py > "$O/engines_100k.json" 2>&1 || exit 1
timeout -k 10 300 python scripts/engine_compare.py 1000000 5 > "$O/engines_1m.json" 2>&1 || exit 1
timeout -k 10 300 python scripts/shard_overhead.py > "$O/shard_overhead.json" 2>&1 || exit 1
timeout -k 10 300 python scripts/engine_compare_lidar.py > "$O/engines_lidar_raw.json" 2>&1 || exit 1
timeout -k 10 100 python scripts/event_overhead.py > "$O/event_overhead.txt" 2>&1 || exit 1
python scripts/prof_summary.py "$O" > "$O/summary.txt" 2>&1
echo done a
exit 0
fi
if [ "$part" = "c" ]; then
# normal estimation: round 2's slot minima against the lists (kernel times, then whole calls on three kinds of cloud)
bash scripts/ab_knn_lists.sh > "$O/ab_knn_lists.txt" 2>&1 || exit 1
timeout -k 10 300 python scripts/normals_timing.py > "$O/normals_timing.json" 2> "$O/normals_timing.err" || exit 1
# the ICP loop: unbounded against bounded passes at C3 and at one shard of an 8-way job (12.5k rows)
bash scripts/ab_nn_bounded.sh > "$O/ab_nn_bounded.txt" 2>&1 || exit 1
bash scripts/ab_nn_bounded.sh 100000 12500 > "$O/ab_nn_bounded_shard.txt" 2>&1 || exit 1
rm -rf gpurun_out/nnb_0 gpurun_out/nnb_1 gpurun_out/knnl_0 gpurun_out/knnl_1
timeout -k 10 300 python scripts/fuzz_bounded.py 3000 > "$O/fuzz_bounded.txt" 2>&1 || exit 1
timeout -k 10 700 python scripts/fuzz_engines.py 4000 31000 > "$O/fuzz_engines.txt" 2>&1 || exit 1
echo done c
exit 0
fi
# C5 at stream length: 200 synthetic frames as KITTI .bin, file -> pose, with the oracle loop beside it
timeout -k 10 600 python scripts/run_sequence.py --make-synthetic /tmp/drive200 --frames 0:200 --oracle > "$O/sequence_200.json" 2> "$O/sequence_200.err" || exit 1
timeout -k 10 300 python scripts/run_sequence.py --data_dir /tmp/drive200 --frames 0:200 --no-prefetch > "$O/sequence_200_noprefetch.json" 2> /dev/null || exit 1
# ... and with the map side of every frame (world points + occupancy grid), the cell set compared with the oracle's
timeout -k 10 600 python scripts/run_sequence.py --data_dir /tmp/drive200 --frames 0:200 --oracle --map > "$O/sequence_200_map.json" 2> "$O/sequence_200_map.err" || exit 1
# same-box A/B of the fused finish + step + transform kernel, and the clock inside the coarse kernel
timeout -k 10 300 python scripts/ab_fuse_finish.py > "$O/ab_fuse_finish.json" 2> "$O/ab_fuse_finish.err" || exit 1
timeout -k 10 300 python scripts/coarse_clock.py 100000 3 > "$O/coarse_clock.json" 2> "$O/coarse_clock.err" || exit 1
timeout -k 10 300 python scripts/ab_r1_r2.py > "$O/ab_r1_r2.json" 2> "$O/ab_r1_r2.err" || true   # needs scripts/ab_r1_libicp.so (round 1's library, built from git archive 0fd41fd)
timeout -k 10 300 python scripts/threshold_sweep.py > "$O/threshold_sweep.json" 2> "$O/threshold_sweep.err" || exit 1
# round 3: the small-cloud kernel against the general path (same box, alternating child processes), its phase clocks,
# the loop-closure verifications side by side, and the iteration-count sensitivity of the oracle loop (CPU)
timeout -k 10 400 python scripts/ab_small.py > "$O/ab_small.json" 2> "$O/ab_small.err" || exit 1
timeout -k 10 300 python scripts/small_clock.py 8000 > "$O/small_clock.json" 2> "$O/small_clock.err" || exit 1
timeout -k 10 300 python scripts/batch_timing.py > "$O/batch_timing.json" 2> "$O/batch_timing.err" || exit 1
timeout -k 10 900 python scripts/iteration_sensitivity.py > "$O/iteration_sensitivity.json" 2> "$O/iteration_sensitivity.err" || exit 1
ICPMI_STREAM_STATS=1 ICPMI_PREFETCH_STATS=1 timeout -k 10 300 python scripts/run_sequence.py --data_dir /tmp/drive200 --frames 0:200 > /dev/null 2> "$O/stream_threads.txt" || true
# the small-cloud regime: kernel time against wall per frame
mkdir -p "$O/small"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/small/stats" -- python3 "$GRAFT_REPO_ROOT/scripts/run_sequence.py" --data_dir /tmp/drive200 --frames 0:40 > "$O/small/sequence_40_under_rocprof.json" 2> "$O/small/stats.err") || exit 1
timeout -k 10 300 python scripts/run_sequence.py --data_dir /tmp/drive200 --frames 0:40 > "$O/small/sequence_40.json" 2> "$O/small/sequence_40.err" || exit 1
python scripts/prof_summary.py "$O" > "$O/summary.txt" 2>&1
echo done
