#!/usr/bin/env python
"""Golden vectors for `minimap2-nd --step 2 --mode 0` (the cns_align command of nextDenovo:356-366 with the re-alignment switched
off): two small files of corrected-read-like sequences with numeric names (incl. reads cut out of longer ones, so that contained
verdicts occur) and the `.ovl` / `.bl` files the compiled reference (oracle/_ref/minimap2-nd) writes for them.
Run in the build container (needs oracle/_ref):  python tests/golden/make_step2_golden.py
"""
import gzip
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import mm_util as M  # noqa: E402
import refpipe  # noqa: E402
from nextdenovo_amd import synth  # noqa: E402

OUT = os.path.join(HERE, "step2")
CASES = [  # (tag, argv between `--step 2 --mode 0` and the files)
    ("ont", ("--dual=yes", "-x", "ava-ont", "-k", "17", "-w", "17", "--minlen", "1000", "--maxhan1", "2000")),
    ("pb", ("--dual=yes", "-x", "ava-pb", "-k", "17", "-w", "10", "--minlen", "700", "--maxhan1", "1500", "--maxhan2", "300")),
    ("ont.I", ("--dual=yes", "-I", "100k", "-x", "ava-ont", "-k", "17", "-w", "17", "--minlen", "1000", "--maxhan1", "2000")),
    ("hifi.self", ("-x", "ava-hifi", "--minlen", "1500")),   # one file against itself (the i == j command), k 51 HPC sketch
]


# The same commands as nextDenovo writes them (nextDenovo:361-364: no --mode, i.e. --mode 2, minimap2/options.c:56): the marked
# candidates of every read are mapped again with the short k-mer sketch.  `deep`: 150x reads of a 12 kb genome -- most reads have
# 200 and more candidates, which the reference re-aligns in batches of --cn targets instead of one by one.
CASES_M2 = [(tag + ".m2", argv) for tag, argv in CASES] + [
    ("deep.m2", ("--dual=yes", "-x", "ava-ont", "-k", "17", "-w", "17", "--minlen", "1000", "--maxhan1", "2000")),
]


# --mode 1: the re-alignment with its own thresholds (20 candidates instead of 200 between its two forms, --cn 50; main.c:455-457)
CASES_M1 = [("ont.m1", dict(CASES)["ont"]), ("deep.m1", dict(CASES_M2)["deep.m2"])]


# --mode 1 with mappings of more than 100,000 anchors: mm_chain_dp_nextdenovo thins the anchors of crowded target positions first
# (minimap2/chain.c:185-226).  Reads across an array of 237 copies of a 36-base unit: against a one-read index every minimizer of the
# array has ~237 occurrences (below -f 1000, which keeps the array out of the first pass, where it has ~2,000), so a candidate read
# brings ~237 x 237 x 4 anchors.  The fixture DEPENDS on the thinning: without it the oracle's records differ
# (tests/test_overlap_oracle.py::test_oracle_step2_anchor_thinning).
CASES_THIN = [("tandem.m1", ("--dual=yes", "-x", "ava-ont", "-k", "17", "-w", "17", "--minlen", "1000", "--maxhan1", "2000", "-f", "1000"))]


# -f FLOAT,INT: reads that chained nothing below the first occurrence threshold are seeded and chained again below the second
# (minimap2/map.c:553-575; in the re-alignment's mappings too, which get the same options: map.c:1045-1113)
CASES_RECHAIN = [("ont.rechain", dict(CASES)["ont"] + ("-f", "5,200")), ("ont.rechain.m2", dict(CASES)["ont"] + ("-f", "5,200")),
                 ("deep.rechain.m1", dict(CASES_M2)["deep.m2"] + ("-f", "60,2000"))]


def files_of(tag):
    if tag.startswith("tandem"):
        return ["tandem.fa.gz", "tandem.fa.gz"]
    if tag.startswith("deep"):
        return ["c.fa.gz", "c.fa.gz"]
    return ["a.fa.gz", "a.fa.gz"] if ".self" in tag else ["a.fa.gz", "b.fa.gz", "a.fa.gz"]


def make_tandem():
    rng = np.random.default_rng(2)
    ul, copies = int(rng.integers(18, 40)), int(rng.integers(215, 300))
    unit = rng.integers(0, 4, ul, dtype=np.uint8)
    left, right = rng.integers(0, 4, 9000, dtype=np.uint8), rng.integers(0, 4, 9000, dtype=np.uint8)
    g = np.concatenate([left] + [unit] * copies + [right])
    rs = synth.simulate_reads(g, 3.5, "hifi", seed=3, mu=9.3, sigma=0.2, min_len=8000)
    with gzip.GzipFile(os.path.join(OUT, "tandem.fa.gz"), "wb", mtime=0) as f:
        for i, sq in enumerate(rs.seqs):
            f.write(b">%d %d 0.99\n%s\n" % (i + 1, sq.size, synth.codes_to_ascii(sq)))
    for tag, argv in CASES_THIN:
        out = os.path.join(OUT, tag + ".ovl")
        refpipe.run([os.path.join(M.REFDIR, "minimap2-nd"), "--step", "2", "--mode", "1", "-t", "3", *argv,
                     *[os.path.join(OUT, f) for f in files_of(tag)], "-o", out])
        print(tag, ul, copies, len(rs.seqs), os.path.getsize(out), os.path.getsize(out + ".bl"))


def make_rechain():
    for tag, argv in CASES_RECHAIN:
        mode = () if tag.endswith(".m2") else ("--mode", "1") if tag.endswith(".m1") else ("--mode", "0")
        out = os.path.join(OUT, tag + ".ovl")
        refpipe.run([os.path.join(M.REFDIR, "minimap2-nd"), "--step", "2", *mode, "-t", "3", *argv,
                     *[os.path.join(OUT, f) for f in files_of(tag)], "-o", out])
        first = argv[:-1] + (argv[-1].partition(",")[0],)   # (the first threshold alone: the fixture must differ from it)
        refpipe.run([os.path.join(M.REFDIR, "minimap2-nd"), "--step", "2", *mode, "-t", "3", *first,
                     *[os.path.join(OUT, f) for f in files_of(tag)], "-o", "/tmp/nd_first.ovl"])
        same = open(out, "rb").read() == open("/tmp/nd_first.ovl", "rb").read()
        print(tag, os.path.getsize(out), os.path.getsize(out + ".bl"), "same as without re-chaining" if same else "re-chaining changes it")


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "tandem":   # (only the anchor-thinning fixture)
        return make_tandem()
    if len(sys.argv) > 1 and sys.argv[1] == "rechain":  # (only the -f FLOAT,INT fixtures)
        return make_rechain()
    g = synth.make_genome(26000, seed=61, n_repeats=2, repeat_len=1200)
    rs = synth.simulate_reads(g, 22, "hifi", seed=62, mu=8.3, sigma=0.35, min_len=2200)
    seqs = list(rs.seqs)
    rng = np.random.default_rng(5)
    for t in range(14):
        a = int(rng.integers(0, len(seqs)))
        if seqs[a].size > 3000:
            s0 = int(rng.integers(0, seqs[a].size - 2400))
            seqs.append(seqs[a][s0:s0 + 2400].copy())
    half = len(seqs) // 2
    for name, lo, hi in (("a.fa.gz", 0, half), ("b.fa.gz", half, len(seqs))):
        with gzip.GzipFile(os.path.join(OUT, name), "wb", mtime=0) as f:
            for i in range(lo, hi):
                f.write(b">%d %d 0.99\n%s\n" % (i + 1, seqs[i].size, synth.codes_to_ascii(seqs[i])))
    gd = synth.make_genome(12000, seed=71, n_repeats=0)
    rd = synth.simulate_reads(gd, 150, "hifi", seed=72, mu=8.0, sigma=0.3, min_len=2200)
    with gzip.GzipFile(os.path.join(OUT, "c.fa.gz"), "wb", mtime=0) as f:
        for i, sq in enumerate(rd.seqs):
            f.write(b">%d %d 0.99\n%s\n" % (i + 1, sq.size, synth.codes_to_ascii(sq)))
    for cases, mode in ((CASES, ("--mode", "0")), (CASES_M2, ()), (CASES_M1, ("--mode", "1"))):
        for tag, argv in cases:
            out = os.path.join(OUT, tag + ".ovl")
            refpipe.run([os.path.join(M.REFDIR, "minimap2-nd"), "--step", "2", *mode, "-t", "3", *argv,
                         *[os.path.join(OUT, f) for f in files_of(tag)], "-o", out])
            print(tag, os.path.getsize(out), os.path.getsize(out + ".bl"))
    make_rechain()
    make_tandem()


if __name__ == "__main__":
    main()
