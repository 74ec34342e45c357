#!/usr/bin/env python
"""Random piles and random parameters against the compiled reference's nextCorrect() (oracle/_ref/nextcorrect.so):
    python tools/fuzz_consensus.py SEED N_SETS
Every set: a random genome / depth / error profile (ONT, CLR, HiFi), its piles (analytic overlaps), random `read_type`, `-fast`, `-split`,
`max_lq_length`, `min_len_aln`, `max_cov_aln`, `min_cov_base`, `min_error_corrected_ratio`; the resident-DB batch entry (ndgpu_correct_piles:
all kernels incl. K12) against the reference pile by pile: length, float32 identity bits, sequence.  On a GPU box, or with NDGPU_SIMT=1 under
the kernel interpreter (a minute or two per set)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refpipe  # noqa: E402
from nextdenovo_amd import api, synth  # noqa: E402


def _reference(args):
    """nextCorrect() of the compiled reference on the piles of one set -- in a process of its own: it writes past its `lq[10]` and
    `score[40]` stack arrays on piles with many low-quality regions (lib/nextcorrect.c:1397-1399, :941-945) and then aborts."""
    inputs, P, max_lq = args
    ref = refpipe.ref_cns()
    out = []
    for seqs, st, en, mal in inputs:
        mlq = min(en[0] // 2, max_lq)   # lib/nextcorrect.py:137: max_lq_length is capped by half the seed
        ln, ide, seq = refpipe.call_nextcorrect(ref, seqs, st, en, mal, P["min_len_aln"], P["max_cov_aln"], P["min_cov_base"], mlq,
                                                P["min_error_corrected_ratio"], P["split"], P["fast"], P["read_type"])
        out.append((ln, float(ide), seq))
    return out


def main():
    from concurrent.futures import ProcessPoolExecutor
    from concurrent.futures.process import BrokenProcessPool
    import multiprocessing as mp
    seed, n_sets = int(sys.argv[1]), int(sys.argv[2])
    if os.environ.get("NDGPU_SIMT"):
        sys.path.insert(0, os.path.join(ROOT, "tests", "simt"))
        import build_simt
        os.environ.setdefault("NDGPU_CONTEXTS", "2")
        api._LIB = api._bind(C.CDLL(os.environ.get("NDGPU_SIMT_LIB") or build_simt.build()))   # (NDGPU_SIMT_LIB: tools/kernel_candidate.py)
    rng = np.random.default_rng(seed)
    bad = total = 0
    for it in range(n_sets):
        prof = str(rng.choice(["ont", "ont", "clr", "hifi"]))
        read_type = {"ont": 1, "clr": int(rng.choice([1, 2])), "hifi": 3}[prof]
        g = synth.make_genome(int(rng.integers(15000, 40000)), seed=int(rng.integers(1, 10 ** 6)), n_repeats=int(rng.integers(0, 3)), repeat_len=1200)
        depth = float(rng.uniform(14, 45))
        kw = dict(mu=9.0, sigma=0.3, min_len=2500) if prof == "hifi" else dict(mu=float(rng.uniform(8.2, 9.0)), sigma=float(rng.uniform(0.3, 0.6)))
        if os.environ.get("FUZZ_HEAVY") and prof != "hifi":   # long seeds (segmented scoring), spliced-in junk (wide bands, low-quality regions)
            g = synth.make_genome(int(rng.integers(60000, 120000)), seed=int(rng.integers(1, 10 ** 6)), n_repeats=int(rng.integers(0, 4)), repeat_len=2500)
            depth, kw = float(rng.uniform(18, 32)), dict(mu=float(rng.uniform(9.8, 10.4)), sigma=0.5)
        rs = synth.simulate_reads(g, depth, prof, seed=int(rng.integers(1, 10 ** 6)), **kw)
        if os.environ.get("FUZZ_HEAVY"):
            for i in range(len(rs.seqs)):
                if rng.random() < 0.2 and rs.seqs[i].size > 4000:
                    # junk replaces a stretch of the same length, so that the analytic overlap coordinates of the read still hold
                    q = int(rng.integers(1000, rs.seqs[i].size - 1500))
                    ln = int(rng.integers(80, 400))
                    rs.seqs[i] = rs.seqs[i].copy()
                    rs.seqs[i][q:q + ln] = rng.integers(0, 4, ln).astype(np.uint8)
        P = dict(min_len_aln=int(rng.choice([300, 500, 1000])), max_cov_aln=int(rng.choice([20, 45, 130])), min_cov_base=int(rng.choice([2, 4, 6])),
                 min_error_corrected_ratio=float(rng.choice([0.6, 0.8, 0.95])), split=int(rng.random() < 0.3), fast=int(rng.random() < 0.25),
                 read_type=read_type)
        max_lq = int(rng.choice([500, 1000, 10000]))
        piles = synth.build_piles(rs, seed_cutoff=1000, max_cov_aln=P["max_cov_aln"], min_len_aln=P["min_len_aln"])
        if not piles:
            continue
        if len(piles) > 14:
            piles = [piles[i] for i in sorted(rng.choice(len(piles), 14, replace=False))]
        words, off, lens = synth.pack_db(rs)
        db = api.ReadDB(words, off, lens)
        recs, poff = synth.flatten_piles(piles)
        t0 = time.time()
        got = db.correct_piles(recs, poff, max_lq_length=max_lq, host_threads=4, **P)
        db.close()
        n_bad = 0
        if os.environ.get("FUZZ_DEVICE_ONLY"):   # (does a crash belong to the device path or to the reference?)
            print(it, "device ran", prof, "piles", len(piles), [int(x[0]) for x in got], flush=True)
            continue
        inputs = [synth.pile_sequences(rs, p) for p in piles]
        try:
            with ProcessPoolExecutor(1, mp_context=mp.get_context("fork")) as ex:
                wants = ex.submit(_reference, (inputs, P, max_lq)).result()
        except BrokenProcessPool:
            print(it, "the reference crashed on this set (device ran: %s)" % [int(x[0]) for x in got], prof, flush=True)
            continue
        for gt, want in zip(got, wants):
            same = gt[0] == want[0] and (want[0] <= 4 or (gt[2] == want[2] and np.float32(gt[1]) == np.float32(want[1])))
            n_bad += not same
        total += len(piles)
        bad += n_bad
        print(it, "equal" if n_bad == 0 else "DIFFER(%d)" % n_bad, prof, "piles", len(piles), P, "max_lq", max_lq, "%.0fs" % (time.time() - t0), flush=True)
    st = api.stats()
    print("piles", total, "mismatches", bad, "| K12 rounds", st["lq_rounds"], "declined", st["lq_declined"], "columns", st["lq_columns"])
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
