#!/bin/bash
# Small-cloud regime (the real caller's: ~7-10k points per frame after the voxel filter): kernel time
# against wall time per call.  Writes N synthetic frames as .bin, runs the stream under rocprofv3.
# usage: scripts/small_prof.sh <tag> [frames]
cd "$GRAFT_REPO_ROOT"
tag=$1; N=${2:-40}
O="$GRAFT_REPO_ROOT/gpurun_out/$tag"
mkdir -p "$O"
python - <<PY
import sys; sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import run_sequence
run_sequence.write_synthetic_drive("/tmp/drive_small", 0, $N)
PY
python scripts/run_sequence.py --data_dir /tmp/drive_small > "$O/sequence.json" 2> "$O/sequence.err" || { tail -5 "$O/sequence.err"; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$GRAFT_REPO_ROOT/scripts/run_sequence.py" --data_dir /tmp/drive_small > "$O/stats.log" 2>&1 || { tail -5 "$O/stats.log"; exit 1; }
cd "$GRAFT_REPO_ROOT"
python scripts/prof_summary.py "$O/stats" > "$O/summary.txt" 2>&1
cat "$O/sequence.json"; head -30 "$O/summary.txt"
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
calls = sum(int(r["Calls"]) for r in rows)
print("kernel time total %.2f ms in %d launches (warm-up of 3 frames included)" % (tot, calls))
PY
