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


def files_of(tag):
    return ["a.fa.gz", "a.fa.gz"] if tag.endswith("self") else ["a.fa.gz", "b.fa.gz", "a.fa.gz"]


def main():
    os.makedirs(OUT, exist_ok=True)
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
    for tag, argv in CASES:
        out = os.path.join(OUT, tag + ".ovl")
        refpipe.run([os.path.join(M.REFDIR, "minimap2-nd"), "--step", "2", "--mode", "0", "-t", "3", *argv,
                     *[os.path.join(OUT, f) for f in files_of(tag)], "-o", out])
        print(tag, os.path.getsize(out), os.path.getsize(out + ".bl"))


if __name__ == "__main__":
    main()
