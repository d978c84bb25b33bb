#!/usr/bin/env python3
"""One registration call as a timeline, from a rocprofv3 kernel trace (CSV): every launch of the LAST call in the trace with
its start, duration and the gap to the launch before it, then the call's parts -- target pre-pass, normals, row order,
iterations (kernel time per iteration) -- summed.  The call is found by its first kernel (k_bbox_partial of the target
pre-pass; a prepared target starts at the rows' k_transform / k_icp_small).
usage: python scripts/call_timeline.py <dir with *_kernel_trace.csv> [--json]"""
import csv
import glob
import json
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("icpmi::", "")
# the calls: a call of this library starts with the target's bounding box (k_bbox_partial) after a stretch without launches
starts = [i for i, r in enumerate(rows) if ("k_bbox_partial" in r["Kernel_Name"] or "k_bbox_single" in r["Kernel_Name"]) and
          (i == 0 or int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 20000 or "k_finish_step" in rows[i - 1]["Kernel_Name"]
           or "copyBuffer" in rows[i - 1]["Kernel_Name"] or "fillBuffer" in rows[i - 1]["Kernel_Name"])]
# (the state's upload may sit between the previous call's copies and this call's first kernel)
starts = [i for k, i in enumerate(starts) if k == 0 or i - starts[k - 1] > 8]
i0 = starts[-1]
call = rows[i0:]
t0 = int(call[0]["Start_Timestamp"])
out, prev_end = [], t0
parts = {"target pre-pass": 0.0, "normals": 0.0, "row order": 0.0, "iterations": 0.0, "other": 0.0}
phase, seen_normals, iters = "target pre-pass", False, 0
for r in call:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = name(r)
    if any(k in n for k in ("k_knn", "k_nn_coarse_rows", "k_normals_from_knn", "k_nn_coarse_groups<true")):
        phase, seen_normals = "normals", True
    elif any(k in n for k in ("k_icp_small", "k_nn_coarse_bounded", "k_nn_coarse_groups<false", "k_nn_resolve", "k_finish", "k_step", "k_nn_prebound1", "k_transform")):
        phase = "iterations"
    elif seen_normals and phase == "normals" and ("k_bbox" in n or "k_morton" in n or "rocprim" in n):
        phase = "row order"
    if n.startswith("k_icp_small") or "k_nn_coarse_bounded" in n or "k_nn_coarse_groups<false" in n:
        iters += 1
    parts[phase] += (e - s) / 1e3
    out.append({"at_us": (s - t0) / 1e3, "us": (e - s) / 1e3, "gap_us": (s - prev_end) / 1e3, "kernel": n[:60], "part": phase})
    prev_end = e
span = (prev_end - t0) / 1e3
busy = sum(o["us"] for o in out)
summary = {"launches": len(out), "span_us": span, "kernel_us": busy, "idle_us": span - busy, "passes": iters,
           "parts_kernel_us": parts, "iterations_kernel_us_per_pass": parts["iterations"] / max(1, iters)}
if "--json" in sys.argv:
    print(json.dumps({"summary": summary, "launches": out}))
else:
    for o in out:
        print("%9.1f us  dur %7.1f  gap %6.1f  %-16s %s" % (o["at_us"], o["us"], o["gap_us"], o["part"], o["kernel"]))
    print(json.dumps(summary, indent=1))
