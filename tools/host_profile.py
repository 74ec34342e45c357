#!/usr/bin/env python
"""Where the host side of the consensus goes (no GPU needed): runs config-2-like piles through the product's host engine with
the oracle backend (tests/csrc/libndhost_test.so) and prints the CPU seconds inside PileEngine::advance by phase."""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402
from nextdenovo_amd import synth  # noqa: E402

so = os.path.join(ROOT, "tests", "csrc", "libndhost_test.so")
csrc = os.path.join(ROOT, "nextdenovo_amd", "csrc")
srcs = [os.path.join(ROOT, "tests", "csrc", "host_harness.cpp")] + [os.path.join(csrc, f) for f in ("consensus.cpp", "poa.cpp", "readdb.cpp")] + \
    [os.path.join(ROOT, "oracle", f) for f in ("ond_oracle.c", "msa_oracle.c")]
subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-o", so] + srcs, check=True)
h = C.CDLL(so)
fn, fr = util.bind_correct(h, "ndtest_correct", "ndtest_free")
g = synth.make_genome(int(float(sys.argv[1]) if len(sys.argv) > 1 else 200000), seed=42, n_repeats=0)
rs = synth.simulate_reads(g, 50, "ont", seed=43)
piles = sorted(synth.build_piles(rs, seed_cutoff=1000), key=lambda p: -int(p["recs"][0][3]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 6]
prof = (C.c_double * 4)()
ex = (C.c_double * 3)()
ext = [0.0] * 3
tot = [0.0] * 4
bases = 0
t0 = time.perf_counter()
for p in piles:
    seqs, st, en, mal = synth.pile_sequences(rs, p)
    ln, ide, _ = util.call_correct(fn, fr, dict(seqs=seqs, aln_start=st, aln_end=en, max_aln=mal, max_lq=min(en[0] // 2, 10000), read_type=1,
                                                fast=0, split=0))
    h.ndtest_advance_profile(prof)
    print("seed %6d bases, %3d reads -> %6d corrected | advance: main %.3f extract %.3f lq1 %.3f lq2+splice %.3f s"
          % (en[0] + 1, len(seqs), ln, prof[0], prof[1], prof[2], prof[3]))
    h.ndtest_extract_profile(ex)
    for i in range(3):
        ext[i] += ex[i]
    for i in range(4):
        tot[i] += prof[i]
    bases += ln
print("after extract: ranking %.3f  POA %.3f  LQ round 1 layout %.3f s" % tuple(ext))
print("total %.1f s wall (oracle alignments included); advance by phase: %s; %.2f us of advance per corrected base"
      % (time.perf_counter() - t0, " ".join("%.3f" % x for x in tot), sum(tot) / max(1, bases) * 1e6))
