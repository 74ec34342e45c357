"""BASELINE configs 4 and 5 at test size inside `pytest -m gpu`: a CLR / `ava-pb` chain at 60x (deep piles, homopolymer-compressed
overlaps, -max_lq_length 1000) and an ultra-long ONT chain with seeds beyond 500 kb, the device's records against the compiled
reference (oracle/_ref/nextcorrect.so) -- what `bench.py --config 4 / 5` checks in its parity block, small enough for the suite."""
import os
from multiprocessing import get_context

import numpy as np
import pytest

import chain_util
import refpipe
import util

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not refpipe.have_ref("nextcorrect.so"), reason="compiled reference did not travel")]


def _ref_worker(item):
    import ctypes as C
    seqs, st, en, mal, mlq, read_type = item
    lib = C.CDLL(os.path.join(refpipe.REFDIR, "nextcorrect.so"))
    fn, fr = util.bind_correct(lib)
    return chain_util.digest(util.call_correct(fn, fr, dict(seqs=seqs, aln_start=st, aln_end=en, max_aln=mal, max_lq=mlq,
                                                             read_type=read_type, fast=0, split=0)))


def _compare(rs, preset, read_type, max_lq, pick_fn, k=40):
    from nextdenovo_amd import api, synth
    sub, off, seeds, _n_bl, (words, word_off, lens) = chain_util.device_piles(rs, preset, k)
    slen = sub[off[:-1].astype(np.int64), 3].astype(np.int64) + 1
    depth = np.diff(off.astype(np.int64))
    pick = pick_fn(slen, depth)
    db = api.ReadDB(words, word_off, lens)
    api.reset_stats()
    res = db.correct_piles(sub, off, read_type=read_type, max_lq_length=max_lq)   # the whole batch, as the stage runs it
    st = api.stats()
    db.close()
    items = []
    for i in pick:
        seqs, s, e, mal = synth.pile_sequences(rs, {"seed": int(seeds[i]), "recs": sub[int(off[i]):int(off[i + 1])]})
        items.append((seqs, s, e, mal, min(e[0] // 2, max_lq), read_type))
    order = sorted(range(len(items)), key=lambda j: -len(items[j][0][0]))
    with get_context("fork").Pool(min(os.cpu_count() or 1, 64)) as pool:
        got = pool.map(_ref_worker, [items[j] for j in order], chunksize=1)
    want = {pick[j]: g for j, g in zip(order, got)}
    bad = [int(i) for i in pick if chain_util.digest(res[i]) != want[i]]
    assert not bad, (len(bad), bad[:5])
    return slen, depth, pick, want, st


def test_clr_deep_piles_match_compiled_reference():
    """60x PacBio-CLR-profile reads (config 4's read model), `ava-pb` overlaps, ovl_sort -k 40, -r clr -max_lq_length 1000."""
    from nextdenovo_amd import synth
    cfg = synth.CONFIGS[4]
    g = synth.make_genome(250000, seed=404, n_repeats=3)
    rs = synth.simulate_reads(g, cfg["depth"], "clr", seed=405, mu=cfg["mu"], sigma=cfg["sigma"], max_len=cfg["max_len"])

    def pick(slen, depth):
        deep = np.argsort(-depth, kind="stable")[:24].tolist()
        return sorted(set(deep) | set(range(0, slen.size, max(1, slen.size // 200))))
    slen, depth, picked, want, st = _compare(rs, cfg["preset"], 2, cfg["max_lq"], pick)
    assert len(picked) >= 150 and depth.max() >= 100                     # deep piles (admission stops at 1.5 x max_cov_aln)
    assert sum(1 for w in want.values() if w[0] > 4) >= 0.9 * len(want)   # corrected, not error seeds


def test_ultralong_seeds_match_compiled_reference():
    """Ultra-long ONT reads (config 5's read model, <= 1 Mb): the three piles with the longest seeds (>= 500 kb) + a sample; the
    alignments of such a pile run to 10^5 edit steps and its scoring DP to hundreds of segments."""
    from nextdenovo_amd import synth
    cfg = synth.CONFIGS[5]
    g = synth.make_genome(1600000, seed=505, n_repeats=5)
    rs = synth.simulate_reads(g, 24.0, "ont", seed=506, mu=11.6, sigma=1.0, max_len=cfg["max_len"])

    def pick(slen, depth):
        longest = [int(i) for i in np.argsort(-slen, kind="stable")[:3] if slen[i] >= 500000]   # (the reference needs ~100 GB of
        return sorted(set(longest) | set(range(0, slen.size, max(1, slen.size // 24))))         # address space for each of them)
    slen, depth, picked, want, st = _compare(rs, cfg["preset"], 1, cfg["max_lq"], pick)
    assert (slen[picked] >= 500000).sum() >= 2, slen.max()
    assert st["score_segments"] > 1000
