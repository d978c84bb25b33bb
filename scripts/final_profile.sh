#!/bin/bash
# Round-end measurement set, run on the GPU box: bench line, kernel stats of the same command,
# PMC passes (one counter group per pass), secondary workloads.  Output under gpurun_out/final/.
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/final"
mkdir -p "$O"
set -x
timeout -k 10 300 python bench.py > "$O/bench.json" 2> "$O/bench.err" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline > "$O/stats.log" 2>&1 || exit 1
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
    tag=$(echo "$grp" | cut -d' ' -f1 | tr 'A-Z' 'a-z')
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$O/pmc_$tag" -- python3 "$GRAFT_REPO_ROOT/scripts/run_align_once.py" 0 100000 30 2 > "$O/pmc_$tag.log" 2>&1 || exit 1
done
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python scripts/bench_extra.py --c4 > "$O/bench_extra.json" 2> "$O/bench_extra.err" || exit 1
timeout -k 10 200 python scripts/engine_compare.py > "$O/engines_100k.json" 2>&1 || exit 1
timeout -k 10 300 python scripts/engine_compare.py 1000000 5 > "$O/engines_1m.json" 2>&1 || exit 1
timeout -k 10 300 python scripts/shard_overhead.py > "$O/shard_overhead.json" 2>&1 || exit 1
timeout -k 10 300 python scripts/engine_compare_lidar.py > "$O/engines_lidar_raw.json" 2>&1 || exit 1
python scripts/prof_summary.py "$O" > "$O/summary.txt" 2>&1
echo done
