#!/bin/bash
# A/B of the 256-query coarse units on few-unit launches (ICPMI_COARSE_HALF_UNITS=<units>; 0 = never), run on the
# GPU box: the 200-frame drive file -> pose, and single registrations of filtered frames.
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/half"
mkdir -p "$O"
timeout -k 10 300 python scripts/run_sequence.py --make-synthetic /tmp/drive200 --frames 0:200 > /dev/null 2>&1 || exit 1
for rep in 1 2; do
for lim in 0 256 512 1024 4096; do
    ICPMI_COARSE_HALF_UNITS=$lim timeout -k 10 300 python scripts/run_sequence.py --data_dir /tmp/drive200 --frames 0:200 > "$O/seq_$lim.json" 2> /dev/null || exit 1
    python - "$O/seq_$lim.json" $lim <<'PY'
import json, sys
q = json.load(open(sys.argv[1]))
print("limit", sys.argv[2], "ms/frame", round(q["gpu_ms_per_frame_file_to_pose"], 4), "median", round(q["gpu_frame_ms_median"], 4), "iters", q["iterations_total"], "ate", q["ate_gpu_m"])
PY
done
done
for lim in 0 256 1024 4096 100000000; do
    echo "limit $lim"; ICPMI_COARSE_HALF_UNITS=$lim timeout -k 10 300 python scripts/shard_overhead.py 2> /dev/null | python -c "
import sys, json
t = sys.stdin.read(); d = json.loads(t[t.index('{'):])
print({k: v['loop_us_per_pass'] for k, v in d.items()})"
done
