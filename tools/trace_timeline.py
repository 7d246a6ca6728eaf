#!/usr/bin/env python3
"""Prints the kernel timeline of the last repetitions in a rocprofv3 --kernel-trace CSV: start and end of each launch in
microseconds relative to the first of the window, so that overlap between streams shows.
usage: trace_timeline.py <kernel_trace.csv> [launches_from_end=40]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%9.1f %9.1f %8.1f us  q%-3s %s" % (s, e, e - s, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
