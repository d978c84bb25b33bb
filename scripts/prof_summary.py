#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats or pmc counter rows) into a short table."""
import signal
signal.signal(signal.SIGPIPE, signal.SIG_DFL)  # `| head` closes the pipe early: end quietly
import csv, glob, sys, collections
d = sys.argv[1]
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats", f)
    for r in csv.DictReader(open(f)):
        print("%-60s calls %5s avg %10.1f us total %10.1f us %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    print("== counters", f)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            print("%-50s %-28s n=%4d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))

import sqlite3
for f in glob.glob(d + "/**/*_results.db", recursive=True):
    print("== kernel stats (rocpd)", f)
    db = sqlite3.connect(f)
    for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        print("%-60s calls %5d avg %10.1f us total %10.1f us %5.2f%%" % (name[:60], calls, avg, total, pct))
