#!/bin/bash
# Kernel-time table of the small-cloud regime (40 frames of the synthetic drive, file -> pose), on the GPU box.
# usage: scripts/small_prof.sh <tag>   (further environment, e.g. ICPMI_COARSE_HALF_UNITS=0, is inherited)
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/$1"
mkdir -p "$O"
[ -d /tmp/drive200 ] || timeout -k 10 300 python scripts/run_sequence.py --make-synthetic /tmp/drive200 --frames 0:200 > /dev/null 2>&1 || exit 1
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$GRAFT_REPO_ROOT/scripts/run_sequence.py" --data_dir /tmp/drive200 --frames 0:40 > "$O/seq.json" 2> "$O/stats.err") || exit 1
python scripts/prof_summary.py "$O/stats" > "$O/summary.txt" 2>&1
head -12 "$O/summary.txt" | cut -c1-150
