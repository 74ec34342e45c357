#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --kernel-trace --pmc passes (rocpd sqlite):
usage: rocprof_pmc_summary.py fetch.db write.db > profiles/<name>.txt
FETCH_SIZE / WRITE_SIZE are in KB (MI355X_MICROARCH.md); values are summed over dispatches per kernel."""
import sqlite3
import sys
from collections import defaultdict


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tables else None
    out = defaultdict(lambda: [0, 0.0])
    if view:
        cols = [r[1] for r in c.execute("pragma table_info(%s)" % view)]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        for name, cname, val in c.execute("select %s, counter_name, value from %s" % (kcol, view)):
            if cname == counter:
                out[name][0] += 1
                out[name][1] += float(val)
    return out


def short(n):
    return n.replace("void ", "").replace("ndgpu::", "").replace("(anonymous namespace)::", "").split("(")[0][:44]


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    print("# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes); KB units")
    print("%-46s %6s %16s %16s %14s" % ("kernel", "calls", "FETCH KB/launch", "WRITE KB/launch", "HBM GB/launch"))
    rows = []
    for k in set(f) | set(w):
        calls = max(f[k][0], w[k][0], 1)
        fk, wk = f[k][1] / calls, w[k][1] / calls
        rows.append(((fk + wk) * calls, short(k), calls, fk, wk))
    for _, k, calls, fk, wk in sorted(rows, reverse=True)[:40]:
        print("%-46s %6d %16.0f %16.0f %14.3f" % (k, calls, fk, wk, (fk + wk) * 1024 / 1e9))


if __name__ == "__main__":
    main()
