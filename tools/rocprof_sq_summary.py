#!/usr/bin/env python
"""Where the wave cycles of every kernel go, from one rocprofv3 --kernel-trace --pmc pass over the SQ counters (rocpd sqlite):
usage: rocprof_sq_summary.py results.db > profiles/<name>.txt
SQ_WAIT_ANY (wavefront parked: s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY (issuing) ~ SQ_WAVE_CYCLES
(MI355X_MICROARCH.md, counter table); all in quad-cycles.  A --pmc pass serialises the kernels: this is every kernel ALONE on the device."""
import sqlite3
import sys
from collections import defaultdict


def main():
    c = sqlite3.connect(sys.argv[1])
    tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tables else None
    if not view:
        sys.exit("no counters_collection view in %s" % sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % view)]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    dcol = "dispatch_id" if "dispatch_id" in cols else None
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    q = "select %s, counter_name, value%s from %s" % (kcol, (", " + dcol) if dcol else "", view)
    for row in c.execute(q):
        name, cname, val = row[0], row[1], row[2]
        acc[name][cname] += float(val)
        if dcol:
            calls[name].add(row[3])

    def short(n):
        return n.replace("void ", "").replace("ndgpu::", "").replace("(anonymous namespace)::", "").split("(")[0][:40]
    print("# %s: every kernel alone on the device (a --pmc pass serialises them); quad-cycle counters summed over the launches" % sys.argv[1])
    print("%-40s %6s %12s %7s %7s %7s %7s %9s %8s" % ("kernel", "calls", "wave_Mcyc", "parked%", "stall%", "issue%", "valu%", "valu/wave", "waves"))
    rows = []
    for k, v in acc.items():
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        rows.append((wc, k, v))
    for wc, k, v in sorted(rows, reverse=True)[:24]:
        waves = v.get("SQ_WAVES", 0.0)
        print("%-40s %6d %12.1f %7.1f %7.1f %7.1f %7.1f %9.0f %8.0f" % (
            short(k), len(calls[k]) if calls[k] else 0, wc / 1e6, 100 * v.get("SQ_WAIT_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc,
            100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * v.get("SQ_ACTIVE_INST_VALU", 0) / wc,
            v.get("SQ_INSTS_VALU", 0) / waves if waves else 0, waves / max(1, len(calls[k]) if calls[k] else 1)))


if __name__ == "__main__":
    main()
