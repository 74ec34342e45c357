#!/usr/bin/env python
"""Golden vectors for `minimap2-nd --step 1 -c` (base-level alignment through the chains, minimap2/align.c:857-913): the `.ovl`
files the compiled reference (oracle/_ref/minimap2-nd) writes
  * for the read sets of make_overlap_golden.py (tests/golden/overlap/{seed,part}.2bit), and
  * for a read set with rearranged reads (tests/golden/cigar/{sv,svq}.2bit): random insertions of 500-900 bases, inverted
    stretches of 700-1100 bases and both in one read, so that gap alignments z-drop, chains are split (mm_split_reg), the
    inversion test fires (mm_test_zdrop, align.c:71-87) and pieces of twice-split chains get their inversion aligned
    (mm_align1_inv, align.c:790-845).
Run in the build container (needs oracle/_ref):  python tests/golden/make_cigar_golden.py
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import mm_util as M  # noqa: E402
from nextdenovo_amd import synth  # noqa: E402

OVL = os.path.join(HERE, "overlap")
OUT = os.path.join(HERE, "cigar")
CASES_C = [  # (file tag, preset, target, query, dual, extra argv); sets "seed"/"part" live in tests/golden/overlap
    ("ont.sxp.dual.c", "ava-ont", "seed", "part", True, ("-c",)),
    ("ont.sxs.c", "ava-ont", "seed", "seed", False, ("-c",)),
    ("pb.sxs.c", "ava-pb", "seed", "seed", False, ("-c",)),
    ("pb.sxp.dual.c", "ava-pb", "seed", "part", True, ("-c",)),
    ("ont.sv.c", "ava-ont", "sv", "svq", True, ("-c",)),
    ("ont.svself.c", "ava-ont", "sv", "sv", False, ("-c",)),
    ("pb.sv.dvt.c", "ava-pb", "sv", "svq", True, ("-c", "--dvt")),
    ("ont.sv.z200.c", "ava-ont", "sv", "svq", True, ("-c", "-z", "200,100", "-s", "120")),
    ("ont.sv.I150k.c", "ava-ont", "sv", "svq", True, ("-c", "-I", "150k")),   # a multi-part index: through the command line only
    # one gap piece: the reference aligns with ksw_extz2_sse (minimap2/align.c:313-331)
    ("ont.sv.O4E2.c", "ava-ont", "sv", "svq", True, ("-c", "-O", "4", "-E", "2")),
    # -f FLOAT,INT: the reads that chained nothing below 6 occurrences are seeded and chained again below 300 (map.c:553-575)
    ("ont.sxp.dual.f6r300.c", "ava-ont", "seed", "part", True, ("-c", "-f", "6,300")),
]


def set_path(name):
    return os.path.join(OVL if name in ("seed", "part") else OUT, name + ".2bit")


def rearranged_reads():
    rng = np.random.default_rng(77)
    g = synth.make_genome(40000, seed=21, n_repeats=4, repeat_len=1500)
    rs = synth.simulate_reads(g, 20, "ont", seed=22, mu=8.9, sigma=0.4, min_len=3000)
    seqs = []
    for n, s in enumerate(rs.seqs):
        s = s.copy()
        kind = n % 4  # 0: as it is, 1: insertion, 2: inversion, 3: insertion then inversion further on
        if kind in (1, 3) and s.size > 4000:
            p = int(s.size * (0.3 if kind == 3 else 0.5))
            s = np.concatenate([s[:p], rng.integers(0, 4, int(rng.integers(500, 900))).astype(np.uint8), s[p:]])
        if kind in (2, 3) and s.size > 5000:
            p = int(s.size * (0.65 if kind == 3 else 0.45))
            ln = int(rng.integers(700, 1100))
            if p + ln < s.size - 500:
                s[p:p + ln] = synth.revcomp_codes(s[p:p + ln])
        seqs.append(s)
    return seqs


def main():
    os.makedirs(OUT, exist_ok=True)
    wd = tempfile.mkdtemp(prefix="ndcig")
    sv, svq = M.dump_reads(wd, [synth.codes_to_ascii(s) for s in rearranged_reads()], seed_cutoff=7000)
    shutil.copy(sv, os.path.join(OUT, "sv.2bit"))
    shutil.copy(svq, os.path.join(OUT, "svq.2bit"))
    for tag, preset, t, q, dual, extra in CASES_C:
        b = M.ref_step1(set_path(t), set_path(q), os.path.join(OUT, tag + ".ovl"), preset, dual, extra)
        print(tag, len(b), "bytes")


if __name__ == "__main__":
    main()
