#!/bin/bash
# A/B of a variant build of the library (scripts/with_lib.py) on the 200-frame drive, on the GPU box.
# usage: scripts/ab_variant.sh <variant.so>
cd "$GRAFT_REPO_ROOT"
V=$1
O="$GRAFT_REPO_ROOT/gpurun_out/variant"
mkdir -p "$O"
timeout -k 10 300 python scripts/run_sequence.py --make-synthetic /tmp/drive200 --frames 0:200 > /dev/null 2>&1 || exit 1
timeout -k 10 500 python scripts/with_lib.py $V -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -m gpu -x -q > "$O/pytest.log" 2>&1; tail -2 "$O/pytest.log"
for rep in 1 2 3; do
    timeout -k 10 300 python scripts/run_sequence.py --data_dir /tmp/drive200 --frames 0:200 > "$O/seq_base.json" 2> /dev/null || exit 1
    timeout -k 10 300 python scripts/with_lib.py $V scripts/run_sequence.py --data_dir /tmp/drive200 --frames 0:200 > "$O/seq_var.json" 2> /dev/null || exit 1
    python - "$O" <<'PY'
import json, sys
for t in ("base", "var"):
    q = json.load(open(sys.argv[1] + "/seq_%s.json" % t))
    print(t, "ms/frame", round(q["gpu_ms_per_frame_file_to_pose"], 4), "median", round(q["gpu_frame_ms_median"], 4), "iters", q["iterations_total"], "ate", q["ate_gpu_m"])
PY
done
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$GRAFT_REPO_ROOT/scripts/with_lib.py" "$GRAFT_REPO_ROOT/$V" "$GRAFT_REPO_ROOT/scripts/run_sequence.py" --data_dir /tmp/drive200 --frames 0:40 > "$O/seq40.json" 2> "$O/stats.err") || exit 1
python scripts/prof_summary.py "$O/stats" > "$O/summary.txt" 2>&1
grep "k_nn_coarse" "$O/summary.txt" | cut -c1-150
