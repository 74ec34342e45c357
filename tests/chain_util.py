"""The device chain of one read set, as bench.py times it: overlap (`minimap2-nd --step 1` path) -> `ovl_sort` -> pile
admission (lib/nextcorrect.py:92-143) -> consensus of every pile.  Used by the GPU tests that work at BASELINE config
sizes, and runnable as a script (one JSON line: per-pile length / identity bits / md5 + the runtime's counters) so that a
test can put the same workload through the scoring kernel's forced paths (NDGPU_K10_* are read once per process)."""
from __future__ import annotations

import hashlib
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def make_set(genome_size, depth, profile="ont", seed=42, mu=9.55, sigma=0.75):
    from nextdenovo_amd import synth
    g = synth.make_genome(int(genome_size), seed=seed)
    return synth.simulate_reads(g, depth, profile, seed=seed + 1, mu=mu, sigma=sigma)


def device_piles(rs, preset="ava-ont", k=40):
    """(recs, pile_off, seeds, n_blacklisted, db arrays): what the consensus stage gets from the stages before it."""
    from nextdenovo_amd import overlap, synth
    words, word_off, lens = synth.pack_db(rs)
    dset = overlap.ReadSet(np.arange(len(rs), dtype=np.uint32), lens, words, word_off)
    with overlap.Index(overlap.preset(preset), dset) as ix:
        raw = ix.map(dset, ix.mid_occ())
    srt, bl, _ = overlap.sort_overlaps([raw], lens.astype(np.uint32), int(lens.min()), k, 300)
    sub, off, seeds = overlap.assemble_piles(srt, lens.size, 500, 500, 130, 10, [i for i, _ in bl])
    return sub, off, seeds, len(bl), (words, word_off, lens)


def digest(rec):
    ln, ide, seq = rec
    if ln <= 4:
        return (int(ln), 0, "")
    return (int(ln), struct.unpack("<I", struct.pack("<f", ide))[0], hashlib.md5(seq).hexdigest())


def run(genome_size, depth, profile="ont", read_type=1, max_piles=0):
    from nextdenovo_amd import api
    rs = make_set(genome_size, depth, profile)
    sub, off, seeds, n_bl, (words, word_off, lens) = device_piles(rs, "ava-ont" if profile == "ont" else "ava-pb")
    if max_piles and seeds.size > max_piles:  # the longest seeds + an even sample of the rest
        order = np.argsort(-(sub[off[:-1].astype(np.int64), 3].astype(np.int64)), kind="stable")
        keep = np.sort(np.concatenate([order[:max_piles // 4], order[max_piles // 4::max(1, (seeds.size - max_piles // 4) // (max_piles - max_piles // 4))]])[:max_piles])
        parts = [sub[int(off[i]):int(off[i + 1])] for i in keep]
        off = np.zeros(len(parts) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([p.shape[0] for p in parts])
        sub = np.ascontiguousarray(np.concatenate(parts))
        seeds = seeds[keep]
    db = api.ReadDB(words, word_off, lens)
    api.reset_stats()
    res = db.correct_piles(sub, off, read_type=read_type)
    st = api.stats()
    db.close()
    return rs, sub, off, seeds, res, st


if __name__ == "__main__":
    gs, depth = float(sys.argv[1]), float(sys.argv[2])
    max_piles = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rs, sub, off, seeds, res, st = run(gs, depth, max_piles=max_piles)
    print(json.dumps({"seeds": [int(s) for s in seeds], "digests": [digest(r) for r in res],
                      "stats": {k: st[k] for k in ("piles", "score_segments", "score_repairs", "score_slow_piles", "cells_msa", "links", "forward_ms",
                                                    "traceback_ms", "lq_rounds", "lq_declined", "lq_ms", "tb_tasks", "tb_walkers", "tb_fallbacks")}}))
