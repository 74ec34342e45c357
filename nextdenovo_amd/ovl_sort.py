#!/usr/bin/env python
"""`ovl_sort` on the MI355X: sort / filter the step-1 overlaps of one seed file into `sorted.ovl` + `.bl`.

Takes the command line nextDenovo writes for the sort_align subtasks (reference nextDenovo:348-350,
util/ovl_sort.c:1040-1078):

    python -m nextdenovo_amd.ovl_sort -m 40g -t 8 -k 40 -i .input.seed.001.idx -o input.seed.001.sorted.ovl input.fofn

`-m`, `-t`, `-d` shape the reference's external merge sort (sorted runs in temporary files when the buffers of `-m` are smaller
than the data, merged at the end: util/ovl_sort.c:1079-1110; the result does not depend on them) and are accepted and ignored: the
records of the input files are decoded into host memory and the device decides by itself.  When raw records, flags and candidates
fit the device (about 360 bytes per raw record: 0.7 G records on a 288 GB MI355X) the sort is one pass; otherwise the raw records
pass the device twice in pieces and the seeds are sorted and filtered in consecutive seed-id ranges (`ndgpu_ovl_sort`,
csrc/ovlsort_engine.hip: sort_out_of_core) -- same records, same order, same `.bl`.  (Cutting the SEED FILE into id ranges by hand is
not equivalent: whether a record is looked at depends on how many earlier records of its file missed the seed table, so the table
must stay whole; the out-of-core form keeps it whole.)  Equal (seed, match, span) keys keep input order, which is what the
reference produces when its buffers are not spilled.  `-H` selects the high-quality-read variant of the filter (`ndgpu_ovl_sort_hq`); `-l 0` is refused.
"""
from __future__ import annotations

import argparse
import sys

import numpy as np

from . import overlap, ovl


def read_idx(path):
    """`.idx` -> (seed_len indexed by read id, shortest seed), util/ovl_sort.c:106-131."""
    ids, lens = [], []
    with open(path) as f:
        for line in f:
            p = line.split("\t")
            if len(p) >= 3:
                ids.append(int(p[0]))
                lens.append(int(p[2]))
    n = max(ids) + 1 if ids else 0
    sl = np.zeros(n, dtype=np.uint32)
    first = {}
    for i, l in zip(ids, lens):
        if i not in first:  # kh_put keeps the first entry of a duplicated id
            first[i] = l
            sl[i] = l
    return sl, (min(first.values()) if first else 0)


def read_fofn(path):
    out = []
    with open(path) as f:
        for line in f:
            if len(line) > 1 and not line.startswith("#"):
                out.append(line.rstrip("\n"))
    return out


def run(argv) -> int:
    ap = argparse.ArgumentParser(prog="ovl_sort", add_help=True)
    ap.add_argument("-i", dest="idx", required=True)
    ap.add_argument("-H", dest="hq", action="store_true")
    ap.add_argument("-m", dest="mem", default="40g")
    ap.add_argument("-t", dest="threads", type=int, default=8)
    ap.add_argument("-k", dest="k", type=int, default=40)
    ap.add_argument("-l", dest="flank", type=int, default=300)
    ap.add_argument("-o", dest="out", required=True)
    ap.add_argument("-d", dest="tmpdir", default=None)
    ap.add_argument("fofn")
    a = ap.parse_args(argv)
    if a.flank <= 0:
        raise SystemExit("[ERROR] -l must be > 0")
    seed_len, min_len = read_idx(a.idx)
    files = [overlap.from_decoded(ovl.decode_ovl(p)) for p in read_fofn(a.fofn)]
    recs, bl, _ = overlap.sort_overlaps(files, seed_len, min_len, a.k, a.flank, hq=a.hq)
    with open(a.out, "wb") as f:
        f.write(overlap.encode(recs, np.zeros(2, dtype=np.uint32)))
    with open(a.out + ".bl", "w") as f:
        for i, k in bl:
            f.write("%d %s\n" % (i, k))
    return 0


if __name__ == "__main__":
    sys.exit(run(sys.argv[1:]))
