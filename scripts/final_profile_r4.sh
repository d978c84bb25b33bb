#!/bin/bash
# Round 4's measurement set, run on the GPU box (two gpurun calls of <= 20 min).  Output under gpurun_out/final4/; the
# summaries are copied into profiles/r4_final/ by scripts/collect_profiles_r4.py (see profiles/README.md).
# usage: scripts/final_profile_r4.sh a   bench lines, kernel stats of the same command, counter passes for both engines' dominant
#                                        kernels, secondary workloads, the culled kernel's in-kernel stamps, C2 at 20k
#        scripts/final_profile_r4.sh b   engine thresholds, offsets, frame stream, small-cloud A/B, batch
#        scripts/final_profile_r4.sh c   the all-pairs kernel's in-kernel clock, the small-cloud regime under rocprofv3, fuzz runs
#        scripts/final_profile_r4.sh d   bench lines, kernel stats, shard shapes, C2 / C3 timelines, stream, small-cloud A/B, fuzz, soak: the part that
#                                        moves with every late change, once more (no counter passes)
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/final4"
mkdir -p "$O"
part=${1:-a}
set -x
if [ "$part" = "a" ]; then
timeout -k 10 400 python bench.py > "$O/bench.json" 2> "$O/bench.err" || exit 1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > "$O/bench_20.json" 2> "$O/bench_20.err" || exit 1
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-traffic > "$O/stats.log" 2>&1) || exit 1
for eng in 0 2; do
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
    tag=$(echo "$grp" | cut -d' ' -f1 | tr 'A-Z' 'a-z')
    (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$O/pmc${eng}_$tag" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" $eng 100000 30 2 > "$O/pmc${eng}_$tag.log" 2>&1) || exit 1
done
done
timeout -k 10 600 python scripts/bench_extra.py --c4 > "$O/bench_extra.json" 2> "$O/bench_extra.err" || exit 1
timeout -k 10 300 python scripts/engine_compare.py 1000000 5 > "$O/engines_1m.json" 2>&1 || exit 1
timeout -k 10 300 python scripts/engine_compare.py 4000000 2 > "$O/engines_4m.json" 2>&1 || exit 1
timeout -k 10 300 python scripts/shard_overhead.py 0 > "$O/shard_overhead_default.json" 2>&1 || exit 1
timeout -k 10 300 python scripts/shard_overhead.py 2 > "$O/shard_overhead_all_pairs.json" 2>&1 || exit 1
timeout -k 10 200 python scripts/groups_clock.py 100000 12 > "$O/groups_clock.json" 2> "$O/groups_clock.err" || exit 1
bash scripts/prof_c2.sh final4/c2_20k 0 0.3 > "$O/c2_20k.log" 2>&1 || exit 1
bash scripts/quick_prof.sh final4/c3_default 0 100000 20 3 > /dev/null 2>&1 || exit 1
python scripts/call_timeline.py "$O/c3_default/stats" > "$O/c3_default_timeline.txt" 2>&1
python scripts/prof_summary.py "$O" > "$O/summary.txt" 2>&1
echo done a
exit 0
fi
if [ "$part" = "b" ]; then
timeout -k 10 200 python scripts/engine_threshold.py > "$O/engine_threshold.json" 2> "$O/engine_threshold.err" || exit 1
timeout -k 10 200 python scripts/offset_timing.py > "$O/offset_timing.json" 2> "$O/offset_timing.err" || exit 1
# C5 at stream length: 200 synthetic frames as KITTI .bin, file -> pose, with the oracle loop beside it; the map side
timeout -k 10 600 python scripts/run_sequence.py --make-synthetic /tmp/drive200 --frames 0:200 --oracle > "$O/sequence_200.json" 2> "$O/sequence_200.err" || exit 1
timeout -k 10 600 python scripts/run_sequence.py --data_dir /tmp/drive200 --frames 0:200 --oracle --map > "$O/sequence_200_map.json" 2> "$O/sequence_200_map.err" || exit 1
timeout -k 10 400 python scripts/ab_small.py > "$O/ab_small.json" 2> "$O/ab_small.err" || exit 1
timeout -k 10 300 python scripts/batch_timing.py > "$O/batch_timing.json" 2> "$O/batch_timing.err" || exit 1
echo done b
exit 0
fi
if [ "$part" = "d" ]; then # the lines that move with every late change, once more on the last code (no counter passes)
timeout -k 10 400 python bench.py > "$O/bench.json" 2> "$O/bench.err" || exit 1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > "$O/bench_20.json" 2> "$O/bench_20.err" || exit 1
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-traffic > "$O/stats.log" 2>&1) || exit 1
timeout -k 10 300 python scripts/shard_overhead.py 0 > "$O/shard_overhead_default.json" 2>&1 || exit 1
timeout -k 10 300 python scripts/shard_overhead.py 2 > "$O/shard_overhead_all_pairs.json" 2>&1 || exit 1
bash scripts/prof_c2.sh final4/c2_20k 0 0.3 > "$O/c2_20k.log" 2>&1 || exit 1
bash scripts/quick_prof.sh final4/c3_default 0 100000 20 3 > /dev/null 2>&1 || exit 1
python scripts/call_timeline.py "$O/c3_default/stats" > "$O/c3_default_timeline.txt" 2>&1
timeout -k 10 600 python scripts/bench_extra.py --c4 > "$O/bench_extra.json" 2> "$O/bench_extra.err" || exit 1
[ -d /tmp/drive200 ] || timeout -k 10 300 python scripts/run_sequence.py --make-synthetic /tmp/drive200 --frames 0:200 > /dev/null 2>&1 || exit 1
timeout -k 10 600 python scripts/run_sequence.py --data_dir /tmp/drive200 --frames 0:200 --oracle > "$O/sequence_200.json" 2> "$O/sequence_200.err" || exit 1
timeout -k 10 400 python scripts/ab_small.py > "$O/ab_small.json" 2> "$O/ab_small.err" || exit 1
timeout -k 10 300 python scripts/fuzz_stopping.py 400 52000 > "$O/fuzz_stopping.txt" 2>&1 || exit 1
timeout -k 10 500 python scripts/fuzz_engines.py 2000 410000 > "$O/fuzz_engines.txt" 2>&1 || exit 1
timeout -k 10 300 python scripts/fuzz_bounded.py 2500 61000 > "$O/fuzz_bounded.txt" 2>&1 || exit 1
timeout -k 10 300 python scripts/soak_stream.py 120 > "$O/soak_stream.txt" 2>&1 || exit 1
python scripts/prof_summary.py "$O" > "$O/summary.txt" 2>&1
echo done d
exit 0
fi
# part c
timeout -k 10 300 python scripts/coarse_clock.py 100000 3 > "$O/coarse_clock.json" 2> "$O/coarse_clock.err" || exit 1
mkdir -p "$O/small"
[ -d /tmp/drive200 ] || timeout -k 10 300 python scripts/run_sequence.py --make-synthetic /tmp/drive200 --frames 0:200 > /dev/null 2>&1 || exit 1
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/small/stats" -- python3 "$GRAFT_REPO_ROOT/scripts/run_sequence.py" --data_dir /tmp/drive200 --frames 0:40 > "$O/small/sequence_40_under_rocprof.json" 2> "$O/small/stats.err") || exit 1
timeout -k 10 300 python scripts/fuzz_stopping.py 400 52000 > "$O/fuzz_stopping.txt" 2>&1 || exit 1
timeout -k 10 500 python scripts/fuzz_engines.py 2000 410000 > "$O/fuzz_engines.txt" 2>&1 || exit 1
timeout -k 10 300 python scripts/fuzz_bounded.py 2500 61000 > "$O/fuzz_bounded.txt" 2>&1 || exit 1
echo done b
