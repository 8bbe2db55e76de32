#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short table (kernel names trimmed).
usage: tools/prof_summary.py <kernel_stats.csv> [out.txt]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)          # drop the argument list
    name = name.replace("void ", "")
    return name if len(name) <= 90 else name[:87] + "..."


rows = list(csv.DictReader(open(sys.argv[1])))
lines = ["%-92s %7s %14s %12s %7s %10s %10s" % ("kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns")]
for r in rows:
    lines.append("%-92s %7s %14s %12.0f %7.2f %10s %10s" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]),
                                                          float(r["Percentage"]), r["MinNs"], r["MaxNs"]))
text = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
else:
    sys.stdout.write(text)
