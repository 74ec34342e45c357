#!/usr/bin/env python
"""Where the non-consensus part of a config-2 bench step goes on the host (GPU box): index build, occurrence threshold, map, sort call,
the copies around it, pile admission -- wall per call, a few repetitions."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from nextdenovo_amd import overlap, synth  # noqa: E402

cfg = synth.CONFIGS[2]
rs = synth.simulate_reads(cfg["genome"](), cfg["depth"], "ont", seed=43, mu=cfg["mu"], sigma=cfg["sigma"], max_len=cfg["max_len"])
words, word_off, lens = synth.pack_db(rs)
ids = np.arange(len(rs), dtype=np.uint32)
dset = overlap.ReadSet(ids, lens, words, word_off)
overlap.words_resident(np.ascontiguousarray(words, dtype=np.uint32))
opt = overlap.preset("ava-ont")
T = {}


def clock(name, f):
    t0 = time.perf_counter()
    r = f()
    T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
    return r


for rep in range(5):
    ix = clock("index_create", lambda: overlap.Index(opt, dset))
    mid = clock("mid_occ", lambda: ix.mid_occ())
    recs = clock("map", lambda: ix.map(dset, mid))
    clock("index_close", lambda: ix.close())
    srt = clock("sort_overlaps", lambda: overlap.sort_overlaps([recs], lens, int(lens.min()), 40, 300))
    clock("assemble_piles", lambda: overlap.assemble_piles(srt[0], int(lens.size), 500, 500, 130, 10, [rid for rid, _ in srt[1]]))
for k, v in T.items():
    print("%-16s %s  (last 3 mean %.1f ms)" % (k, " ".join("%6.1f" % x for x in v), sum(v[-3:]) / 3))
print("records", recs.shape, "sorted", srt[0].shape, "sort stats", srt[2])
