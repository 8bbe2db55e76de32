#!/usr/bin/env python3
"""Prints the per-launch durations (us) of one kernel from a rocprofv3 --kernel-trace CSV, in launch order.
usage: python3 tools/launch_trace.py <dir with *_kernel_trace.csv> <kernel name substring>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
prev_end = None
for s, e in rows:
    gap = "" if prev_end is None else " gap %.1f" % ((s - prev_end) / 1e3)
    print("%.1f%s" % ((e - s) / 1e3, gap))
    prev_end = e
