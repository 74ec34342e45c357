#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db) as text.
usage: rocprof_summary.py results.db > profiles/<name>.txt"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
print("# rocprofv3 --kernel-trace --stats summary of %s" % sys.argv[1])
print("%-60s %8s %14s %14s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    short = name.replace("void ", "").replace("ndgpu::", "").replace("(anonymous namespace)::", "").split("(")[0]
    print("%-60s %8d %14.1f %14.1f %8.2f" % (short[:60], calls, total, avg, pct))
