#!/usr/bin/env python
"""Golden vectors for the overlap path: a small seeded read set (with interspersed and tandem repeats, so
that repetitive-minimizer filtering and equal-coordinate anchors occur) dumped by the reference seq_dump,
and the `.ovl` files the compiled reference `minimap2-nd --step 1` (oracle/_ref/minimap2-nd) writes for it.
Run in the build container (needs oracle/_ref):  python tests/golden/make_overlap_golden.py
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

CASES = [  # (file tag, preset, target, query, dual, extra argv)
    ("ont.sxp.dual", "ava-ont", "seed", "part", True, ()),
    ("ont.sxs", "ava-ont", "seed", "seed", False, ()),
    ("ont.sxp.f002", "ava-ont", "seed", "part", False, ("-f", "0.002")),
    ("ont.sxs.f30", "ava-ont", "seed", "seed", False, ("-f", "30")),
    ("pb.sxp.dual", "ava-pb", "seed", "part", True, ()),
    ("pb.sxs", "ava-pb", "seed", "seed", False, ()),
    ("ont.sxs.I200k", "ava-ont", "seed", "seed", False, ("-I", "200k")),
    ("ont.sxp.dual.I150k", "ava-ont", "seed", "part", True, ("-I", "150k")),
    # HiFi reads (hseed / hpart): k = 51, w = 51, homopolymer-compressed -- the two-word k-mer sketch (sketch.c:283-356)
    ("hifi.sxs", "ava-hifi", "hseed", "hseed", False, ()),
    ("hifi.sxp.dual", "ava-hifi", "hseed", "hpart", True, ()),
    ("hifi.sxs.f40", "ava-hifi", "hseed", "hseed", False, ("-f", "40")),   # nextDenovo passes -f seed_depth*20 (config_parser.py:46-47)
    # -f FLOAT,INT: a read that chained nothing below the first threshold is seeded and chained again below the second (map.c:553-575)
    ("ont.sxp.dual.f6r300", "ava-ont", "seed", "part", True, ("-f", "6,300")),
    ("pb.sxs.f4r60", "ava-pb", "seed", "seed", False, ("-f", "4,60")),
    ("hifi.sxp.f3r50", "ava-hifi", "hseed", "hpart", False, ("-f", "3,50")),
    # chains of ONE anchor pass (-n 1; the thresholds lowered so that they reach the output)
    ("ont.sxp.dual.n1", "ava-ont", "seed", "part", True, ("-n", "1", "-m", "15", "--minlen", "14")),
    # long k-mers outside ava-hifi's 51: even (a k-mer can equal its reverse complement), three words
    ("hifi.sxp.dual.k40", "ava-hifi", "hseed", "hpart", True, ("-k", "40", "-w", "30")),
    ("hifi.sxs.k70", "ava-hifi", "hseed", "hseed", False, ("-k", "70", "-w", "40")),
]
# --mode 3: chain ends trimmed (nd_fix_bad_ends) and every hit extended into the unaligned read ends (nd_extend_ends,
# minimap2/map.c:340-482) before the step-1 filter; with --dvt the extension is skipped for hits that are not near-dovetails
CASES_M3 = [
    ("hifi.sxs.m3", "ava-hifi", "hseed", "hseed", False, ("--mode", "3")),
    ("hifi.sxp.dual.dvt.m3", "ava-hifi", "hseed", "hpart", True, ("--mode", "3", "--dvt")),
    ("ont.sxp.dual.m3", "ava-ont", "seed", "part", True, ("--mode", "3")),
    ("pb.sxs.m3", "ava-pb", "seed", "seed", False, ("--mode", "3")),
]
SETS = ("seed", "part", "hseed", "hpart")


def main():
    out = os.path.join(HERE, "overlap")
    os.makedirs(out, exist_ok=True)
    rng = np.random.default_rng(3)
    g = synth.make_genome(45000, seed=8, n_repeats=6, repeat_len=1800)
    for pos, unit, copies in ((9000, 37, 50), (30000, 151, 14)):
        blk = np.tile(rng.integers(0, 4, unit).astype(np.uint8), copies)
        g[pos:pos + blk.size] = blk
    rs = synth.simulate_reads(g, 22, "ont", seed=9, mu=8.6, sigma=0.5, min_len=800)
    wd = tempfile.mkdtemp(prefix="ndovl")
    seed, part = M.dump_reads(wd, [synth.codes_to_ascii(s) for s in rs.seqs], seed_cutoff=6000)
    files = {"seed": seed, "part": part}
    for k, p in files.items():
        shutil.copy(p, os.path.join(out, k + ".2bit"))
    gh = synth.make_genome(60000, seed=18, n_repeats=5, repeat_len=2500)
    for pos, unit, copies in ((12000, 61, 40),):
        blk = np.tile(rng.integers(0, 4, unit).astype(np.uint8), copies)
        gh[pos:pos + blk.size] = blk
    rh = synth.simulate_reads(gh, 24, "hifi", seed=19, mu=8.9, sigma=0.3, min_len=3000)
    hs, hp = M.dump_reads(os.path.join(wd, "hifi"), [synth.codes_to_ascii(s) for s in rh.seqs], seed_cutoff=8000)
    files["hseed"], files["hpart"] = hs, hp
    for k in ("hseed", "hpart"):
        shutil.copy(files[k], os.path.join(out, k + ".2bit"))
    for tag, preset, t, q, dual, extra in CASES + CASES_M3:
        b = M.ref_step1(files[t], files[q], os.path.join(out, tag + ".ovl"), preset, dual, extra)
        print(tag, len(b), "bytes")


if __name__ == "__main__":
    main()
