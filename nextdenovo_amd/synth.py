"""Seeded synthetic long-read generator + analytic pile builder.

This is workload tooling for bench.py / tests (the reference ships no data:
``test_data/reads_test.fa.gz`` is a missing blob).  It follows the
workload definitions of SURVEY.md section 8(d):

* ``make_genome``   uniform random ACGT (+ optional rRNA-like repeat copies)
* ``simulate_reads`` lognormal read lengths, 50/50 strand, ONT/CLR error
  profile (sub/ins/del, 2x rate inside homopolymers >= 3)
* ``build_piles``   what ``minimap2-nd --step 1 | ovl_sort | nextcorrect.py``
  would hand to ``nextCorrect`` for every seed, derived analytically from
  the true read positions: one overlap record per (seed, read) pair in the
  ``decode_ovl`` field order used by lib/nextcorrect.py:106
  (``t_name, rev, t_s, t_e, q_name, q_s, q_e, match``), sorted like
  util/ovl_sort.c:246-255 (match desc, span asc), admitted with the rules of
  lib/nextcorrect.py:124-126.

Base codes follow the reference 2-bit DB (lib/bseq.c:11-20): A0 C1 G2 T3.
"""
from __future__ import annotations

import numpy as np

PROFILES = {
    # name: (sub, ins, del)  -- SURVEY.md section 8(d) config 2 / config 4
    "ont": (0.03, 0.04, 0.05),
    "clr": (0.015, 0.09, 0.045),
    "hifi": (0.002, 0.003, 0.003),
}

CKPT = 32  # genome->read coordinate checkpoints every CKPT genome bases


def make_genome(size: int, seed: int = 42, n_repeats: int = 7, repeat_len: int = 5000) -> np.ndarray:
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=size, dtype=np.uint8)
    if n_repeats and size > 4 * n_repeats * repeat_len:
        unit = rng.integers(0, 4, size=repeat_len, dtype=np.uint8)
        for pos in rng.integers(0, size - repeat_len, size=n_repeats):
            cp = unit.copy()
            # ~1 % divergence between copies
            m = rng.random(repeat_len) < 0.01
            cp[m] = (cp[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
            g[pos:pos + repeat_len] = cp
    return g


def _homopolymer_mask(seg: np.ndarray) -> np.ndarray:
    """True where the base sits inside a run of >= 3 identical bases."""
    n = seg.size
    if n < 3:
        return np.zeros(n, dtype=bool)
    same = seg[1:] == seg[:-1]
    # run id per position
    start = np.ones(n, dtype=bool)
    start[1:] = ~same
    rid = np.cumsum(start) - 1
    rl = np.bincount(rid)
    return rl[rid] >= 3


def mutate(seg: np.ndarray, rng: np.random.Generator, profile: str, first: int = 0):
    """Apply the error model to a genome segment (codes 0..3).

    Returns (read_codes, ckpt) where ckpt[j] is the read offset that
    corresponds to segment offset first + j*CKPT (monotone).
    """
    sub, ins, dele = PROFILES[profile]
    n = seg.size
    hp = _homopolymer_mask(seg)
    scale = np.where(hp, 2.0, 1.0)
    r = rng.random(n)
    is_del = r < dele * scale
    is_sub = (~is_del) & (r < (dele + sub) * scale)
    n_ins = (rng.random(n) < ins * scale).astype(np.int64)
    # occasional longer insertions
    n_ins += (rng.random(n) < ins * 0.15 * scale).astype(np.int64)
    keep = (~is_del).astype(np.int64)
    cnt = n_ins + keep
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(cnt, out=off[1:])
    total = int(off[-1])
    out = rng.integers(0, 4, size=total, dtype=np.uint8)  # insertion filler
    base = seg.copy()
    ns = int(is_sub.sum())
    if ns:
        base[is_sub] = (base[is_sub] + rng.integers(1, 4, size=ns).astype(np.uint8)) & 3
    kept_idx = np.nonzero(keep)[0]
    out[off[kept_idx] + n_ins[kept_idx]] = base[kept_idx]
    ckpt = off[first::CKPT].astype(np.int64)
    return out, ckpt


def revcomp_codes(a: np.ndarray) -> np.ndarray:
    return (3 - a[::-1]).astype(np.uint8)


class ReadSet:
    """Reads in their sequenced orientation + truth needed to derive overlaps."""

    def __init__(self):
        self.seqs = []      # list[np.ndarray uint8 codes], as sequenced
        self.gstart = []    # genome start of the covered segment
        self.gend = []      # genome end (exclusive)
        self.rev = []       # 1 if the read is the reverse complement of the genome
        self.ckpt = []      # forward-strand read offsets at absolute genome positions that are multiples of CKPT
        self.gfirst = []    # first such genome position inside the read

    def __len__(self):
        return len(self.seqs)

    def total_bases(self):
        return int(sum(s.size for s in self.seqs))


def simulate_reads(genome: np.ndarray, depth: float, profile: str = "ont", seed: int = 43,
                   mu: float = 9.55, sigma: float = 0.75, min_len: int = 1000,
                   max_len: int = 200000) -> ReadSet:
    rng = np.random.default_rng(seed)
    G = genome.size
    rs = ReadSet()
    target = depth * G
    tot = 0
    while tot < target:
        L = int(np.clip(rng.lognormal(mu, sigma), min_len, min(max_len, G)))
        s = int(rng.integers(0, G - L + 1))
        seg = genome[s:s + L]
        gfirst = ((s + CKPT - 1) // CKPT) * CKPT
        codes, ck = mutate(seg, rng, profile, gfirst - s)
        if codes.size < min_len:
            continue
        rev = int(rng.random() < 0.5)
        rs.seqs.append(revcomp_codes(codes) if rev else codes)
        rs.gstart.append(s)
        rs.gend.append(s + L)
        rs.rev.append(rev)
        rs.ckpt.append(ck)
        rs.gfirst.append(gfirst)
        tot += codes.size
    return rs


def build_piles(rs: ReadSet, seed_cutoff: int = 1000, min_ovl: int = 500, max_cov_aln: int = 130,
                min_len_aln: int = 500, min_cov_seed: int = 10, seed_ids=None):
    """Return a list of piles.  pile = dict(seed=id, recs=np.ndarray[n,8] uint32);
    recs[0] is the self record (util/ovl_sort.c:827-835).  Coordinates inclusive.
    Vectorised over the candidate reads of each seed."""
    n = len(rs)
    gs = np.asarray(rs.gstart, dtype=np.int64)
    ge = np.asarray(rs.gend, dtype=np.int64)
    rv = np.asarray(rs.rev, dtype=np.int64)
    lens = np.asarray([s.size for s in rs.seqs], dtype=np.int64)
    ck_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([c.size for c in rs.ckpt], out=ck_off[1:])
    ck_flat = np.concatenate(rs.ckpt) if n else np.zeros(0, dtype=np.int64)
    ck_n = np.diff(ck_off)
    order = np.argsort(gs, kind="stable")
    gs_sorted = gs[order]

    gf = np.asarray(rs.gfirst, dtype=np.int64)

    def fwd(ids, g):  # exact genome->read offsets at absolute CKPT-grid positions
        j = (g - gf[ids]) // CKPT
        j = np.minimum(np.maximum(j, 0), ck_n[ids] - 1)
        return ck_flat[ck_off[ids] + j]

    piles = []
    ids = range(n) if seed_ids is None else seed_ids
    for si in ids:
        L = int(lens[si])
        if L < seed_cutoff:
            continue
        a, b = int(gs[si]), int(ge[si])
        hi = int(np.searchsorted(gs_sorted, b, side="left"))
        cand = order[:hi]
        cand = cand[(ge[cand] > a) & (cand != si)]
        if cand.size == 0:
            continue
        lo_g = np.maximum(a, gs[cand])
        hi_g = np.minimum(b, ge[cand])
        ok = hi_g - lo_g >= min_ovl + 2 * CKPT
        cand, lo_g, hi_g = cand[ok], lo_g[ok], hi_g[ok]
        if cand.size == 0:
            continue
        sid = np.full(cand.size, si, dtype=np.int64)
        # overlap ends = the same absolute grid positions in both reads (like exact
        # minimizer anchors), one grid step inside the true overlap
        g0s = ((lo_g + CKPT - 1) // CKPT + 1) * CKPT
        g1s = (hi_g // CKPT - 1) * CKPT
        ts_f, te_f = fwd(sid, g0s), fwd(sid, g1s)
        qs_f, qe_f = fwd(cand, g0s), fwd(cand, g1s)
        ok = (te_f - ts_f >= min_ovl) & (qe_f - qs_f >= min_ovl)
        cand, ts_f, te_f, qs_f, qe_f = cand[ok], ts_f[ok], te_f[ok] - 1, qs_f[ok], qe_f[ok] - 1
        if cand.size == 0:
            continue
        if rs.rev[si]:
            t_s, t_e = L - 1 - te_f, L - 1 - ts_f
        else:
            t_s, t_e = ts_f, te_f
        Lq = lens[cand]
        qr = rv[cand] == 1
        q_s = np.where(qr, Lq - 1 - qe_f, qs_f)
        q_e = np.where(qr, Lq - 1 - qs_f, qe_f)
        rev = rv[cand] ^ int(rs.rev[si])
        match = (0.45 * (t_e - t_s + 1)).astype(np.int64)
        # util/ovl_sort.c:246-255: match desc, span asc
        o = np.lexsort((t_e - t_s, -match))
        recs = np.stack([np.full(cand.size, si), rev, t_s, t_e, cand, q_s, q_e, match], axis=1)[o]
        # lib/nextcorrect.py:124-126 admission
        span = recs[:, 3] - recs[:, 2]
        keep = np.zeros(recs.shape[0], dtype=bool)
        total = L
        lim = max_cov_aln * 1.5
        for r in range(recs.shape[0]):
            if span[r] < min_len_aln or total / L > lim:
                continue
            keep[r] = True
            total += int(span[r]) + 1
        if total / L < min_cov_seed:
            continue
        out = np.concatenate([np.asarray([[si, 0, 0, L - 1, si, 0, L - 1, 0]], dtype=np.int64), recs[keep]])
        piles.append({"seed": si, "recs": out.astype(np.uint32)})
    return piles


def flatten_piles(piles):
    """(recs[n,8] uint32 contiguous, pile_off[n_piles+1] uint64) for ndgpu_correct_piles."""
    off = np.zeros(len(piles) + 1, dtype=np.uint64)
    np.cumsum([p["recs"].shape[0] for p in piles], out=off[1:])
    recs = np.ascontiguousarray(np.concatenate([p["recs"] for p in piles]).astype(np.uint32)) if piles else \
        np.zeros((0, 8), dtype=np.uint32)
    return recs, off


def pack_db(rs: ReadSet):
    """Reference .2bit payload for every read: (words uint32, word_off uint64, len uint32)."""
    n = len(rs)
    lens = np.asarray([s.size for s in rs.seqs], dtype=np.uint32)
    nw = (lens.astype(np.int64) + 15) // 16
    word_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(nw, out=word_off[1:])
    words = np.zeros(int(word_off[-1]), dtype=np.uint32)
    for i, s in enumerate(rs.seqs):
        words[int(word_off[i]):int(word_off[i + 1])] = pack_2bit_msb(s)
    return words, word_off[:-1].copy(), lens


def pack_2bit_msb(codes: np.ndarray) -> np.ndarray:
    """Reference .2bit word layout (lib/bseq.c:114-139): 16 bases per uint32,
    first base in the two most-significant bits."""
    n = codes.size
    nw = (n + 15) // 16
    pad = np.zeros(nw * 16, dtype=np.uint32)
    pad[:n] = codes
    pad = pad.reshape(nw, 16)
    shifts = (30 - 2 * np.arange(16)).astype(np.uint32)
    return (pad << shifts).sum(axis=1, dtype=np.uint64).astype(np.uint32)


def codes_to_ascii(codes: np.ndarray) -> bytes:
    return np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes()


def pile_sequences(rs: ReadSet, pile) -> tuple[list[bytes], list[int], list[int], int]:
    """What lib/nextcorrect.py:183-199 (worker) builds for nextCorrect():
    ASCII strings via getseq (revcomp applied), aln_start, aln_end, max_aln_length."""
    seqs, st, en = [], [], []
    recs = pile["recs"]
    max_aln = int(recs[0][3]) + 1
    for r in recs:
        t, rev, t_s, t_e, q, q_s, q_e, _ = (int(v) for v in r)
        sub = rs.seqs[q][q_s:q_e + 1]
        if rev:
            sub = revcomp_codes(sub)
        seqs.append(codes_to_ascii(sub))
        st.append(t_s)
        en.append(t_e)
        v = t_e - t_s + q_e - q_s + 2
        if v > max_aln and t != q:
            max_aln = v
    return seqs, st, en, max_aln


# ---- BASELINE configs 3-5: genomes with repeat families, read sets generated on all host cores ----

def make_genome_repeats(size: int, seed: int, families, frac: float) -> np.ndarray:
    """Uniform random ACGT with `frac` of it covered by copies of repeat families.  families = [(unit length range,
    divergence range, weight)]: every copy is its family's unit with substitutions at a rate drawn from the divergence
    range (plus one-base indels at a tenth of it), dropped at a random place."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=size, dtype=np.uint8)
    units = []
    for (lo, hi), div, weight in families:
        n_units = max(1, int(weight))
        for _ in range(n_units):
            units.append((rng.integers(0, 4, size=int(rng.integers(lo, hi + 1)), dtype=np.uint8), div))
    covered, target = 0, int(frac * size)
    while covered < target:
        unit, (d0, d1) = units[int(rng.integers(0, len(units)))]
        cp = unit.copy()
        d = rng.uniform(d0, d1)
        m = rng.random(cp.size) < d
        cp[m] = (cp[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
        drop = rng.random(cp.size) < d * 0.1
        cp = cp[~drop]
        pos = int(rng.integers(0, size - cp.size))
        g[pos:pos + cp.size] = cp
        covered += cp.size
    return g


CONFIGS = {
    # name: genome builder, depth, read profile, lognormal mu / sigma, longest read, overlap preset, max_lq_length, sort -k
    # (BASELINE.json configs[1..4]; lognormal N50 ~ exp(mu + sigma^2); sort -k as lib/config_parser.py:44)
    2: dict(genome=lambda: make_genome(4600000, seed=42), depth=50.0, profile="ont", mu=9.55, sigma=0.75, max_len=200000,
            preset="ava-ont", max_lq=10000, name="synthetic E. coli-like 4.6 Mb, 50x ONT (N50 ~ 20 kb)"),
    3: dict(genome=lambda: make_genome_repeats(140000000, 342, [((1000, 6000), (0.02, 0.08), 400)], 0.20), depth=40.0, profile="ont",
            mu=9.55, sigma=0.75, max_len=200000, preset="ava-ont", max_lq=10000,
            name="synthetic D. melanogaster-like 140 Mb with 20 % interspersed repeats (1-6 kb families, 2-8 % divergence), 40x ONT"),
    4: dict(genome=lambda: make_genome_repeats(120000000, 442, [((1000, 8000), (0.03, 0.10), 300)], 0.12), depth=60.0, profile="clr",
            mu=9.03, sigma=0.6, max_len=100000, preset="ava-pb", max_lq=1000,
            name="synthetic A. thaliana-like 120 Mb, 60x PacBio CLR (N50 ~ 12 kb)"),
    5: dict(genome=lambda: make_genome_repeats(250000000, 542, [((280, 320), (0.05, 0.15), 40), ((900, 6500), (0.02, 0.12), 200),
                                                                 ((2000, 9000), (0.01, 0.06), 60)], 0.45), depth=30.0, profile="ont",
            mu=10.51, sigma=1.0, max_len=1000000, preset="ava-ont", max_lq=10000,
            name="synthetic human chr1-like 250 Mb with 45 % repeats (Alu / L1-style families), 30x ultra-long ONT (N50 ~ 100 kb, <= 1 Mb)"),
}

_MP_GENOME = None


def _mp_chunk(args):
    """One chunk of a read set: its own random stream, the error model of simulate_reads, reads packed on the spot."""
    depth, profile, seed, mu, sigma, min_len, max_len = args
    rng = np.random.default_rng(seed)
    genome = _MP_GENOME
    G = genome.size
    target, tot = depth * G, 0
    seqs, gstart, gend, revs = [], [], [], []
    while tot < target:
        L = int(np.clip(rng.lognormal(mu, sigma), min_len, min(max_len, G)))
        s = int(rng.integers(0, G - L + 1))
        codes, _ck = mutate(genome[s:s + L], rng, profile, 0)
        if codes.size < min_len:
            continue
        rev = int(rng.random() < 0.5)
        seqs.append(revcomp_codes(codes) if rev else codes)
        gstart.append(s)
        gend.append(s + L)
        revs.append(rev)
        tot += codes.size
    words = [pack_2bit_msb(s) for s in seqs]
    return seqs, gstart, gend, revs, words


def simulate_reads_mp(genome: np.ndarray, depth: float, profile: str, seed: int, mu: float, sigma: float, max_len: int,
                      min_len: int = 1000, chunks: int = 96, procs: int = 0):
    """The read set of simulate_reads() generated as `chunks` independent streams (seed + chunk number) on the host's cores:
    the same set whatever the core count.  Returns (ReadSet without the analytic-pile checkpoints, words, word_off, lens)."""
    import multiprocessing as mp
    import os
    global _MP_GENOME
    _MP_GENOME = genome
    from . import hostinfo
    procs = procs or min(chunks, hostinfo.effective_cpus())   # (the CPUs the process can have: a cgroup quota counts)
    jobs = [(depth / chunks, profile, seed * 1000 + c, mu, sigma, min_len, max_len) for c in range(chunks)]
    with mp.get_context("fork").Pool(procs) as pool:
        parts = pool.map(_mp_chunk, jobs, chunksize=1)
    _MP_GENOME = None
    rs = ReadSet()
    wl = []
    for seqs, gs, ge, rv, words in parts:
        rs.seqs += seqs
        rs.gstart += gs
        rs.gend += ge
        rs.rev += rv
        wl += words
    n = len(rs.seqs)
    lens = np.asarray([s.size for s in rs.seqs], dtype=np.uint32)
    word_off = np.zeros(n, dtype=np.uint64)
    if n:
        word_off[1:] = np.cumsum([w.size for w in wl])[:-1]
    return rs, (np.concatenate(wl) if wl else np.zeros(0, dtype=np.uint32)), word_off, lens


def pile_sequences_from_recs(rs: ReadSet, recs):
    """pile_sequences() for a bare record array."""
    return pile_sequences(rs, {"recs": recs})
