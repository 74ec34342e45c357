#!/usr/bin/env python
"""Concurrency summary of a rocprofv3 --kernel-trace result (rocpd sqlite .db): per kernel the summed duration, and for
the whole trace the time during which at least one kernel ran (union), the mean number of kernels in flight, and the
share of wall time per kernel name when time is split evenly between the kernels in flight.
usage: rocprof_timeline.py results.db [t0_frac t1_frac] > profiles/<name>.txt"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
if not rows:
    sys.exit("no kernels")
T0, T1 = min(r[1] for r in rows), max(r[2] for r in rows)
if len(sys.argv) > 3:
    a, b = float(sys.argv[2]), float(sys.argv[3])
    T0, T1 = T0 + (T1 - T0) * a, T0 + (T1 - T0) * b
    rows = [(n, max(s, T0), min(e, T1)) for n, s, e in rows if e > T0 and s < T1]


def short(n):
    return n.replace("void ", "").replace("ndgpu::", "").replace("(anonymous namespace)::", "").split("(")[0][:56]


ev = []
for i, (n, s, e) in enumerate(rows):
    ev.append((s, 1, i))
    ev.append((e, -1, i))
ev.sort()
live = set()
share = defaultdict(float)
tot = defaultdict(float)
calls = defaultdict(int)
union = 0.0
area = 0.0
prev = ev[0][0]
for t, k, i in ev:
    if t > prev and live:
        dt = t - prev
        union += dt
        area += dt * len(live)
        for j in live:
            share[short(rows[j][0])] += dt / len(live)
    prev = t
    if k > 0:
        live.add(i)
    else:
        live.discard(i)
for n, s, e in rows:
    tot[short(n)] += e - s
    calls[short(n)] += 1
wall = T1 - T0
print("# %s: window %.1f ms, some kernel running %.1f ms (%.0f %%), mean kernels in flight while busy %.2f"
      % (sys.argv[1], wall * 1e-6, union * 1e-6, 100.0 * union / wall, area / max(union, 1)))
print("%-58s %7s %12s %12s %7s" % ("kernel", "calls", "sum_ms", "share_ms", "share%"))
for n in sorted(share, key=lambda k: -share[k]):
    print("%-58s %7d %12.2f %12.2f %7.2f" % (n, calls[n], tot[n] * 1e-6, share[n] * 1e-6, 100.0 * share[n] / wall))

# coarse timeline: per bin the share of time with a kernel running, mean kernels in flight, and the kernels that own the bin
BIN = 25e6
nb = int((T1 - T0) / BIN) + 1
busy = [0.0] * nb
conc = [0.0] * nb
own = [defaultdict(float) for _ in range(nb)]
live = set()
prev = ev[0][0]
for t, k, i in ev:
    if t > prev and live:
        a = prev
        while a < t:
            b_i = int((a - T0) / BIN)
            e = min(t, T0 + (b_i + 1) * BIN)
            if 0 <= b_i < nb:
                busy[b_i] += e - a
                conc[b_i] += (e - a) * len(live)
                for j in live:
                    own[b_i][short(rows[j][0])[:22]] += (e - a) / len(live)
            a = e
    prev = t
    if k > 0:
        live.add(i)
    else:
        live.discard(i)
print("\n# timeline, %d ms bins: busy %%, kernels in flight while busy, owners (share of the bin)" % (BIN / 1e6))
for b_i in range(nb):
    top = sorted(own[b_i].items(), key=lambda kv: -kv[1])[:4]
    print("%7.0f ms  %3.0f%%  %4.1f  %s" % (b_i * BIN / 1e6, 100.0 * busy[b_i] / BIN, conc[b_i] / max(busy[b_i], 1),
                                          "  ".join("%s %.0f%%" % (n, 100.0 * v / BIN) for n, v in top)))
