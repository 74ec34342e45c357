#!/usr/bin/env python
"""Overlap engine at workload scale on the GPU box: synthetic read set -> index + all-vs-all map on the MI355X,
optionally the compiled reference (oracle/_ref/minimap2-nd -t N) on the same .2bit for timing and a byte
comparison of the two .ovl files.  Diagnostic tool (bench.py carries the judged numbers)."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nextdenovo_amd import overlap, ovl, synth  # noqa: E402


def make_set(genome_size, depth, profile, seed=42):
    g = synth.make_genome(int(genome_size), seed=seed)
    rs = synth.simulate_reads(g, depth, profile, seed=seed + 1)
    n = len(rs.seqs)
    lens = np.asarray([s.size for s in rs.seqs], dtype=np.uint32)
    words = [synth.pack_2bit_msb(s) for s in rs.seqs]
    woff = np.zeros(n, dtype=np.uint64)
    woff[1:] = np.cumsum([w.size for w in words])[:-1]
    return overlap.ReadSet(np.arange(1, n + 1, dtype=np.uint32), lens, np.concatenate(words), woff)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-size", type=float, default=4.6e6)
    ap.add_argument("--depth", type=float, default=50)
    ap.add_argument("--profile", default="ont")
    ap.add_argument("--dual", action="store_true")
    ap.add_argument("--ref-threads", type=int, default=0, help="also run oracle/_ref/minimap2-nd with this many threads")
    ap.add_argument("--repeat", type=int, default=2)
    a = ap.parse_args()
    preset = "ava-ont" if a.profile == "ont" else "ava-pb"
    t0 = time.time()
    rs = make_set(a.genome_size, a.depth, a.profile)
    print("reads %d bases %d gen %.1fs" % (len(rs), int(rs.lens.sum()), time.time() - t0), flush=True)
    opt = overlap.preset(preset)
    if a.dual:
        opt.no_dual = 0
    res = {}
    blob = None
    for it in range(a.repeat):
        t0 = time.time()
        with overlap.Index(opt, rs) as ix:
            t1 = time.time()
            mid = ix.mid_occ()
            recs = ix.map(rs, mid)
            t2 = time.time()
            blob = overlap.encode(recs, np.zeros(2, dtype=np.uint32))
            t3 = time.time()
            st = ix.stats()
            info = ix.stat()
        res = dict(index_s=t1 - t0, map_s=t2 - t1, encode_s=t3 - t2, total_s=t3 - t0, mid_occ=mid, n_recs=int(recs.size),
                   ovl_bytes=len(blob), index=info, stats=st)
        print(json.dumps(res), flush=True)
    if a.ref_threads:
        wd = tempfile.mkdtemp(prefix="ovlscale")
        p = os.path.join(wd, "reads.2bit")
        ovl.write_2bit(p, rs.ids, rs.lens, rs.words, rs.word_off)
        out = os.path.join(wd, "ref.ovl")
        cmd = [os.path.join(ROOT, "oracle", "_ref", "minimap2-nd"), "--step", "1"] + (["--dual=yes"] if a.dual else []) + \
              ["-t", str(a.ref_threads), "-x", preset, p, p, "-o", out]
        t0 = time.time()
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.time() - t0
        ref = open(out, "rb").read()
        print(json.dumps(dict(ref_threads=a.ref_threads, ref_s=dt, ref_bytes=len(ref), identical=(ref == blob),
                              speedup=dt / res["total_s"])), flush=True)


if __name__ == "__main__":
    main()
