"""BASELINE configs 3, 4 and 5 as the 8-GPU layout runs them, at a genome size the suite can afford: the SAME repeat families
(unit lengths, divergence, copies per family, share of the genome), read model, preset, `ovl_sort -k` and `-max_lq_length` as
`bench.py --config N`, the genome cut to 10-16 Mb; the reads are dealt into 8 seed files (util/seq_dump.c:87-92) and seed file 3 --
what rank 3 of 8 corrects -- goes through `stage.Shard` (its raw_align jobs incl. the mirrors, ovl_sort, pile admission) and the
consensus, whole.  A sample of its piles (every k-th + the longest seeds + the deepest piles) is compared with the compiled
reference (oracle/_ref/nextcorrect.so): length, float32 identity bits, md5 of the bases -- bench.py's parity block."""
import hashlib
import os
import struct
from multiprocessing import get_context

import numpy as np
import pytest

import refpipe
import util

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not refpipe.have_ref("nextcorrect.so"), reason="compiled reference did not travel")]

_CTX = None
SHARD = 3   # the seed file corrected: rank 3 of 8 computes (3, seed t >= 3) itself and needs the mirrors (t < 3, seed 3)


def _ref_worker(i):
    import ctypes as C
    from nextdenovo_amd import synth
    rs, piles, max_lq, read_type = _CTX
    seqs, st, en, mal = synth.pile_sequences(rs, piles[i])
    lib = C.CDLL(os.path.join(refpipe.REFDIR, "nextcorrect.so"))
    fn, fr = util.bind_correct(lib)
    ln, ide, seq = util.call_correct(fn, fr, dict(seqs=seqs, aln_start=st, aln_end=en, max_aln=mal, max_lq=min(en[0] // 2, max_lq),
                                                  read_type=read_type, fast=0, split=0))
    return i, ln, (struct.unpack("<I", struct.pack("<f", ide))[0] if ln > 4 else 0), (hashlib.md5(seq).hexdigest() if ln > 4 else "")


# (config, genome size, families scaled to it: the full-size config's unit count x size / full size keeps the copies per family)
CASES = {
    3: dict(size=12_000_000, seed=342, families=[((1000, 6000), (0.02, 0.08), 34)], frac=0.20, n_sample=40),
    4: dict(size=10_000_000, seed=442, families=[((1000, 8000), (0.03, 0.10), 25)], frac=0.12, n_sample=40),
    5: dict(size=16_000_000, seed=542, families=[((280, 320), (0.05, 0.15), 3), ((900, 6500), (0.02, 0.12), 13), ((2000, 9000), (0.01, 0.06), 4)],
            frac=0.45, n_sample=24, min_piles=100),   # (ultra-long reads: a seed file of 16 Mb at 30x holds ~150 seeds)
}


@pytest.mark.parametrize("config", [3, 4, 5])
def test_one_rank_of_eight_matches_compiled_reference(config):
    global _CTX
    from nextdenovo_amd import api, hostinfo, stage, synth
    cfg, case = synth.CONFIGS[config], dict(CASES[config])
    small = float(os.environ.get("NDGPU_TEST_SHARD_MB", "0"))   # (a dry run under the kernel interpreter: tools/gpu_tests_interpreted.py)
    if small:
        case["size"], case["n_sample"] = int(small * 1e6), 6
    genome = synth.make_genome_repeats(case["size"], case["seed"], case["families"], case["frac"])
    rs, words, word_off, lens = synth.simulate_reads_mp(genome, cfg["depth"], cfg["profile"], 43 + config, cfg["mu"], cfg["sigma"], cfg["max_len"])
    d = int(round(cfg["depth"]))
    sort_k = (d - 2) if d <= 30 else min(d - 5, 40)                                   # lib/config_parser.py:44
    read_type = {"ont": 1, "clr": 2, "hifi": 3}[cfg["profile"]]
    sh = stage.Shard(words, word_off, lens, preset=cfg["preset"], seed_cutoff=1000, read_cutoff=500, n_seed_files=8, sort_k=sort_k)
    try:
        jobs = sh.jobs_of(SHARD)
        assert len(jobs) >= 8 and any(j[1] != SHARD for j in jobs) and any(j[1] == SHARD for j in jobs)   # its own jobs and the mirrors
        sub, off, seeds, n_bl = sh.piles(SHARD)
    finally:
        sh.close()
    n = int(seeds.size)
    assert n >= (case.get("min_piles", 200) if not small else 3), n
    db = api.ReadDB(words, word_off, lens)
    try:
        api.reset_stats()
        res = db.correct_piles(sub, off, read_type=read_type, max_lq_length=cfg["max_lq"])   # the whole shard, as the stage runs it
        st = api.stats()
    finally:
        db.close()
    piles = [{"seed": int(seeds[i]), "recs": sub[int(off[i]):int(off[i + 1])]} for i in range(n)]
    slen = np.asarray([int(p["recs"][0][3]) + 1 for p in piles])
    depth = np.diff(off.astype(np.int64))
    # the reference needs ~ (0.4 x 2 x seed length)^2 bytes of address space per pile: the longest seeds of config 5 are sampled below 400 kb
    ok_len = np.flatnonzero(slen <= 400000)
    pick = set(ok_len[::max(1, ok_len.size // case["n_sample"])].tolist())
    pick |= set(ok_len[np.argsort(-slen[ok_len], kind="stable")[:4]].tolist())
    pick |= set(ok_len[np.argsort(-depth[ok_len], kind="stable")[:4]].tolist())
    pick = sorted(pick, key=lambda i: -int(slen[i]))
    _CTX = (rs, piles, cfg["max_lq"], read_type)
    with get_context("fork").Pool(min(hostinfo.effective_cpus(), 32)) as pool:
        got = dict((g[0], g[1:]) for g in pool.imap_unordered(_ref_worker, pick, chunksize=1))
    _CTX = None
    bad = []
    for i in pick:
        ln, ide, seq = res[i]
        mine = (ln, struct.unpack("<I", struct.pack("<f", ide))[0] if ln > 4 else 0, hashlib.md5(seq).hexdigest() if ln > 4 else "")
        if got[i][0] == 3:      # the reference itself ran out of memory on this seed (lib/nextcorrect.c:2254-2261): nothing to compare
            continue
        full = ln > 4 or got[i][0] > 4   # (error seeds -- 2 uncorrectable, 4 all clipped -- carry the code only)
        if (mine != tuple(got[i])) if full else (mine[0] != got[i][0]):
            bad.append((i, int(seeds[i]), got[i][0], ln))
    assert not bad, (len(bad), bad[:5])
    assert sum(1 for i in pick if got[i][0] > 4) >= 0.8 * len(pick)                  # corrected seeds, not error codes
    assert st["lq_declined"] <= st["lq_rounds"] // 10 and st["piles"] == n
    if small:
        return
    if config == 3:
        assert depth.max() >= 60                                                        # repeat copies pile up (large pile depth)
    if config == 5:
        assert slen.max() >= 250000                                                     # ultra-long seeds in the shard
