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


def mutate(seg: np.ndarray, rng: np.random.Generator, profile: str):
    """Apply the error model to a genome segment (codes 0..3).

    Returns (read_codes, ckpt) where ckpt[j] is the read offset that
    corresponds to genome offset j*CKPT of the segment (monotone).
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
    ckpt = off[::CKPT].astype(np.int64)
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
        self.ckpt = []      # forward-strand genome->read checkpoints

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
        codes, ck = mutate(seg, rng, profile)
        if codes.size < min_len:
            continue
        rev = int(rng.random() < 0.5)
        rs.seqs.append(revcomp_codes(codes) if rev else codes)
        rs.gstart.append(s)
        rs.gend.append(s + L)
        rs.rev.append(rev)
        rs.ckpt.append(ck)
        tot += codes.size
    return rs


def _fwd_pos(rs: ReadSet, i: int, g: int) -> int:
    """Offset in the forward-strand version of read i for genome position g
    (snapped down to the checkpoint grid)."""
    j = (g - rs.gstart[i]) // CKPT
    ck = rs.ckpt[i]
    j = min(max(j, 0), ck.size - 1)
    return int(ck[j])


def build_piles(rs: ReadSet, seed_cutoff: int = 1000, min_ovl: int = 500, max_cov_aln: int = 130,
                min_len_aln: int = 500, min_cov_seed: int = 10, sort_depth: int = 40,
                seed_ids=None):
    """Return a list of piles.  pile = dict(seed=id, recs=np.ndarray[n,8] uint32)
    recs[0] is the self record (ovl_sort.c:827-835).  Coordinates inclusive."""
    n = len(rs)
    gs = np.asarray(rs.gstart)
    ge = np.asarray(rs.gend)
    order = np.argsort(gs, kind="stable")
    gs_sorted = gs[order]
    lens = np.asarray([s.size for s in rs.seqs])
    piles = []
    ids = range(n) if seed_ids is None else seed_ids
    for si in ids:
        L = int(lens[si])
        if L < seed_cutoff:
            continue
        a, b = int(gs[si]), int(ge[si])
        # candidates: reads starting before b and ending after a
        hi = int(np.searchsorted(gs_sorted, b, side="left"))
        cand = order[:hi]
        cand = cand[ge[cand] > a]
        recs = []
        for qi in cand:
            qi = int(qi)
            if qi == si:
                continue
            lo_g = max(a, int(gs[qi]))
            hi_g = min(b, int(ge[qi]))
            if hi_g - lo_g < min_ovl + 2 * CKPT:
                continue
            # snap to checkpoint grids of both reads: use genome positions that are
            # multiples of CKPT relative to BOTH starts -> recompute per read
            g0 = lo_g + CKPT
            g1 = hi_g - CKPT
            ts_f, te_f = _fwd_pos(rs, si, g0), _fwd_pos(rs, si, g1)
            # genome positions actually used by the seed snap
            g0s = rs.gstart[si] + ((g0 - rs.gstart[si]) // CKPT) * CKPT
            g1s = rs.gstart[si] + ((g1 - rs.gstart[si]) // CKPT) * CKPT
            qs_f, qe_f = _fwd_pos(rs, qi, g0s), _fwd_pos(rs, qi, g1s)
            if te_f - ts_f < min_ovl or qe_f - qs_f < min_ovl:
                continue
            te_f -= 1
            qe_f -= 1
            # to seed-read orientation
            if rs.rev[si]:
                t_s, t_e = L - 1 - te_f, L - 1 - ts_f
            else:
                t_s, t_e = ts_f, te_f
            Lq = int(lens[qi])
            if rs.rev[qi]:
                q_s, q_e = Lq - 1 - qe_f, Lq - 1 - qs_f
            else:
                q_s, q_e = qs_f, qe_f
            rev = rs.rev[si] ^ rs.rev[qi]
            match = int(0.45 * (t_e - t_s + 1))
            recs.append((si, rev, t_s, t_e, qi, q_s, q_e, match))
        if not recs:
            continue
        recs.sort(key=lambda r: (-r[7], r[3] - r[2]))
        # lib/nextcorrect.py:124-126 admission
        out = [(si, 0, 0, L - 1, si, 0, L - 1, 0)]
        total = L
        for r in recs:
            if r[3] - r[2] < min_len_aln or total / L > max_cov_aln * 1.5:
                continue
            out.append(r)
            total += r[3] - r[2] + 1
        if total / L < min_cov_seed:
            continue
        piles.append({"seed": si, "recs": np.asarray(out, dtype=np.uint32)})
    return piles


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
