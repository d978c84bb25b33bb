#!/bin/bash
# Same-box comparison of several builds of the library on the 200-frame drive (file -> pose), alternating.
# usage: scripts/ab_libs.sh <tag> <lib.so> [<lib.so> ...]     ("product" = the in-tree library)
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
mkdir -p "$O"
[ -d /tmp/drive200 ] || timeout -k 10 300 python scripts/run_sequence.py --make-synthetic /tmp/drive200 --frames 0:200 > /dev/null 2>&1 || exit 1
for rep in 1 2 3; do
    for lib in "$@"; do
        name=$(basename $lib .so)
        if [ "$lib" = product ]; then
            ICPMI_PREFETCH_STATS=1 timeout -k 10 300 python scripts/run_sequence.py --data_dir /tmp/drive200 --frames 0:200 > "$O/seq_$name.json" 2> "$O/err_$name.txt" || exit 1
        else
            ICPMI_PREFETCH_STATS=1 timeout -k 10 300 python scripts/with_lib.py $lib scripts/run_sequence.py --data_dir /tmp/drive200 --frames 0:200 > "$O/seq_$name.json" 2> "$O/err_$name.txt" || exit 1
        fi
        python - "$O/seq_$name.json" $name <<'PY'
import json, sys
q = json.load(open(sys.argv[1]))
print("%-12s ms/frame %.4f median %.4f p95 %.4f iters %d ate %r" % (sys.argv[2], q["gpu_ms_per_frame_file_to_pose"], q["gpu_frame_ms_median"], q["gpu_frame_ms_p95"], q["iterations_total"], q["ate_gpu_m"]))
PY
        grep prefetch "$O/err_$name.txt" | sed 's/^/             /'
    done
done
