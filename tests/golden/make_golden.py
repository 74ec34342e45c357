#!/usr/bin/env python
"""Generate the committed golden fixtures from the REAL reference.

Run in the build container (needs /root/reference compiled into oracle/_ref by
`make -C oracle ref`):

    python tests/golden/make_golden.py

Writes (all seeded, deterministic):
  tests/golden/align_pairs.npz  query/target pairs + the reference's align()/align_hq()
                                 output (lib/align.c:563-578) as column-kind streams
  tests/golden/piles.npz        seed piles produced by the reference stage chain
                                 (seq_dump -> minimap2-nd --step 1 -> ovl_sort -> pile
                                 assembly of lib/nextcorrect.py:92-143) + the reference's
                                 nextCorrect() result for each (lib/nextcorrect.c:2219)
  tests/golden/poa.npz          inputs/outputs of the reference poa_to_consensus()
                                 (lib/dag.c:658-694)
  tests/golden/stage/           a whole seed_cns stage: .2bit/.idx read DB, sorted.ovl(.bl) from the
                                 reference chain and the cns.fasta(.idx) written by the reference's own
                                 lib/nextcorrect.py -p 1 (default, -s, -b)
The reference ships no golden vectors for this path; these files
are what pins parity on machines where /root/reference does not exist.
"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import refpipe  # noqa: E402
from nextdenovo_amd import synth  # noqa: E402

ASC = np.frombuffer(b"ACGT", dtype=np.uint8)


class Aln(C.Structure):
    _fields_ = [("shift", C.c_uint), ("aln_len", C.c_uint), ("aln_t_s", C.c_uint), ("aln_t_e", C.c_uint),
                ("aln_t_len", C.c_uint), ("aln_q_len", C.c_uint), ("q_aln_str", C.c_char_p),
                ("t_aln_str", C.c_char_p)]


def ref_align(lib, q: bytes, t: bytes, hq: int):
    """Call the reference align()/align_hq() exactly as nextCorrect does
    (lib/nextcorrect.c:2243-2285): zeroed V, triangular D from malloc_vd."""
    total = len(q) + len(t) + 2
    max_mem_d = int(total * 0.4 + 1) if not hq else int(total * 0.5 + 1)
    V = C.POINTER(C.c_int)()
    D = C.c_void_p()
    lib.malloc_vd(C.byref(V), C.byref(D), C.c_uint64(max_mem_d))
    lib.clean_V(V, max_mem_d)
    a = Aln()
    tb, qb = C.create_string_buffer(total + 8), C.create_string_buffer(total + 8)
    a.t_aln_str = C.cast(tb, C.c_char_p)
    a.q_aln_str = C.cast(qb, C.c_char_p)
    a.aln_t_s = 0
    a.aln_len = 0
    (lib.align_hq if hq else lib.align)(q, len(q), t, len(t), C.byref(a), V, D)
    lib.destory_vd(V, D)
    n = a.aln_len
    ts, qs = tb.raw[:n], qb.raw[:n]
    return n, a.aln_t_len, a.aln_q_len, ts, qs


def strings_to_ops(ts: bytes, qs: bytes) -> np.ndarray:
    t = np.frombuffer(ts, dtype=np.uint8)
    q = np.frombuffer(qs, dtype=np.uint8)
    ops = np.zeros(t.size, dtype=np.uint8)
    ops[t == ord("-")] = 1
    ops[q == ord("-")] = 2
    return ops


def make_pairs(lib):
    rng = np.random.default_rng(1234)
    pairs = []

    def noisy(seg, prof, seed):
        return synth.mutate(seg, np.random.default_rng(seed), prof)[0]

    g = rng.integers(0, 4, size=60000, dtype=np.uint8)
    # typical overlaps at several lengths and both profiles
    for i, L in enumerate([40, 120, 600, 900, 1500, 2600, 4000, 5200, 8000, 12000]):
        for prof in ("ont", "clr", "hifi"):
            s = int(rng.integers(0, g.size - L))
            pairs.append((noisy(g[s:s + L], prof, 100 + i), noisy(g[s:s + L], prof, 200 + i), 0))
    for i, L in enumerate([200, 800, 1500, 6000]):  # align_hq thresholds
        s = int(rng.integers(0, g.size - L))
        pairs.append((noisy(g[s:s + L], "hifi", 300 + i), noisy(g[s:s + L], "hifi", 400 + i), 1))
        pairs.append((noisy(g[s:s + L], "ont", 300 + i), noisy(g[s:s + L], "ont", 400 + i), 1))
    # identical, empty, tiny
    a = g[100:700].copy()
    pairs += [(a, a.copy(), 0), (a[:5], a[:5].copy(), 0), (a[:1], a[:1].copy(), 0), (a[:0], a[:0], 0),
              (a[:30], a[:0], 0), (a[:0], a[:30], 0), (a[:3], a[10:14], 0)]
    # unrelated sequences (edit budget exhausted) and partially related ones
    pairs += [(rng.integers(0, 4, 800, dtype=np.uint8), rng.integers(0, 4, 800, dtype=np.uint8), 0),
              (rng.integers(0, 4, 3000, dtype=np.uint8), rng.integers(0, 4, 3300, dtype=np.uint8), 0)]
    # long indels: below and above the 250-column gap limit (lib/align.c:542-545)
    base = g[2000:5000].copy()
    for gap in (100, 249, 251, 252, 300, 600):
        pairs.append((np.concatenate([base[:1500], base[1500 + gap:]]), base.copy(), 0))
        pairs.append((base.copy(), np.concatenate([base[:1500], base[1500 + gap:]]), 0))
    # an unavoidable long gap: flanks without 'A', insert of pure 'A' -> >250 consecutive gap
    # columns in the traceback (aln_len = 2 marker) and a live band beyond 253 diagonals
    X = rng.integers(1, 4, 1500, dtype=np.uint8)
    Y = rng.integers(1, 4, 1600, dtype=np.uint8)
    for gap in (200, 260, 400):
        Z = np.zeros(gap, dtype=np.uint8)
        pairs.append((np.concatenate([X, Y]), np.concatenate([X, Z, Y]), 0))
        pairs.append((np.concatenate([X, Z, Y]), np.concatenate([X, Y]), 0))
    # overhangs at the ends (forced moves, lib/align.c:512)
    pairs.append((np.concatenate([rng.integers(0, 4, 60, dtype=np.uint8), base[:900]]), base[:900].copy(), 0))
    pairs.append((base[:900].copy(), np.concatenate([base[:900], rng.integers(0, 4, 80, dtype=np.uint8)]), 0))
    # low-complexity / tandem repeats: wide live bands (band re-centring, lib/align.c:473-489)
    unit = np.asarray([0, 1], dtype=np.uint8)
    tr = np.tile(unit, 1500)
    pairs.append((noisy(tr, "ont", 500), noisy(tr, "ont", 501), 0))
    pairs.append((np.zeros(2500, dtype=np.uint8), np.zeros(2300, dtype=np.uint8), 0))
    pairs.append((noisy(np.tile(np.asarray([0, 1, 2], dtype=np.uint8), 2500), "clr", 502),
                  noisy(np.tile(np.asarray([0, 1, 2], dtype=np.uint8), 2500), "clr", 503), 0))
    hom = np.zeros(4000, dtype=np.uint8)
    hom[::97] = 1
    pairs.append((noisy(hom, "ont", 504), noisy(hom, "ont", 505), 0))

    rec = {"q": [], "t": [], "hq": [], "aln_len": [], "t_used": [], "q_used": [], "ops": []}
    for q, t, hq in pairs:
        qa, ta = ASC[q].tobytes(), ASC[t].tobytes()
        n, tu, qu, ts, qs = ref_align(lib, qa, ta, hq)
        rec["q"].append(q)
        rec["t"].append(t)
        rec["hq"].append(hq)
        rec["aln_len"].append(n)
        rec["t_used"].append(tu if n else 0)
        rec["q_used"].append(qu if n else 0)
        rec["ops"].append(strings_to_ops(ts, qs) if n > 2 else np.zeros(0, dtype=np.uint8))
    return rec


def ragged(list_of_arrays, dtype=np.uint8):
    off = np.zeros(len(list_of_arrays) + 1, dtype=np.int64)
    np.cumsum([a.size for a in list_of_arrays], out=off[1:])
    flat = np.concatenate(list_of_arrays).astype(dtype) if list_of_arrays else np.zeros(0, dtype=dtype)
    return flat, off


def pack2(codes: np.ndarray) -> np.ndarray:
    """4 bases per byte."""
    n = codes.size
    pad = np.zeros((n + 3) // 4 * 4, dtype=np.uint8)
    pad[:n] = codes
    pad = pad.reshape(-1, 4)
    return (pad[:, 0] | (pad[:, 1] << 2) | (pad[:, 2] << 4) | (pad[:, 3] << 6)).astype(np.uint8)


def make_piles(lib):
    out = {"codes": [], "lens": [], "pile_off": [0], "aln_start": [], "aln_end": [], "max_aln": [], "max_lq": [],
           "read_type": [], "fast": [], "split": [], "exp_len": [], "exp_ide": [], "exp_seq": []}
    code_of = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        code_of[ch] = i
    for prof, preset, rt, gseed in (("ont", "ava-ont", 1, 11), ("clr", "ava-pb", 2, 12), ("hifi", "ava-hifi", 3, 13)):
        g = synth.make_genome(36000, seed=gseed, n_repeats=0)
        if prof == "hifi":
            # two haplotypes (SNPs + small indels + a homopolymer length difference every ~400 bp):
            # exercises the phasing branches of generate_lqseqs_from_tags_kmer (nextcorrect.c:789-898)
            rng = np.random.default_rng(gseed)
            h2 = g.copy()
            for pos in range(300, g.size - 300, 400):
                kind = int(rng.integers(0, 3))
                if kind == 0:
                    h2[pos] = (h2[pos] + 1 + int(rng.integers(0, 3))) & 3
                elif kind == 1:
                    h2 = np.concatenate([h2[:pos], rng.integers(0, 4, int(rng.integers(1, 4)), dtype=np.uint8), h2[pos:]])
                else:
                    h2[pos:pos + 6] = h2[pos]
                    h2 = np.concatenate([h2[:pos], h2[pos:pos + 2], h2[pos:]])
            r1 = synth.simulate_reads(g, 16, prof, seed=gseed + 100, mu=8.2, sigma=0.3, min_len=1000)
            r2 = synth.simulate_reads(h2, 16, prof, seed=gseed + 200, mu=8.2, sigma=0.3, min_len=1000)
            rs = synth.ReadSet()
            rs.seqs = r1.seqs + r2.seqs
        else:
            rs = synth.simulate_reads(g, 32, prof, seed=gseed + 100, mu=8.0, sigma=0.35, min_len=1000)
        wd = tempfile.mkdtemp(prefix="ndgold")
        fa = os.path.join(wd, "reads.fa")
        refpipe.write_fasta(fa, [synth.codes_to_ascii(s) for s in rs.seqs])
        idxs, so = refpipe.run_overlap_chain(wd, fa, seed_cutoff=2500, preset=preset)
        piles = list(refpipe.read_piles(idxs, so, min_len_seed=1250))
        pick = piles[::max(1, len(piles) // 5)][:5]
        for k, (seed, seqs, st, en, mal, recs) in enumerate(pick):
            variants = [(0, 0)]
            if k == 0:
                variants += [(1, 0), (0, 1)]  # -fast and -s
            for fast, split in variants:
                mlq = min(en[0] // 2, 10000 if prof == "ont" else 1000)
                if prof == "hifi" and (fast or split):
                    continue
                ln, ide, seq = refpipe.call_nextcorrect(lib, seqs, st, en, mal, max_lq_length=mlq, split=split,
                                                        fast=fast, read_type=rt)
                for s in seqs:
                    c = code_of[np.frombuffer(s, dtype=np.uint8)]
                    assert c.max(initial=0) < 4
                    out["codes"].append(pack2(c))
                    out["lens"].append(len(s))
                out["pile_off"].append(out["pile_off"][-1] + len(seqs))
                out["aln_start"] += list(st)
                out["aln_end"] += list(en)
                out["max_aln"].append(mal)
                out["max_lq"].append(mlq)
                out["read_type"].append(rt)
                out["fast"].append(fast)
                out["split"].append(split)
                out["exp_len"].append(ln)
                out["exp_ide"].append(ide)
                out["exp_seq"].append(np.frombuffer(seq or b"", dtype=np.uint8))
    return out


def make_poa(lib):
    """struct seq_ { u16 order, kscore, len; char seq[10000]; } (lib/nextcorrect.h:62-68)"""
    stride = 6 + 10000
    lib.poa_to_consensus.argtypes = [C.c_void_p, C.c_int]
    lib.poa_to_consensus.restype = C.c_void_p
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    rng = np.random.default_rng(77)
    cases = []
    for ci in range(40):
        L = int(rng.integers(9, 400))
        n = int(rng.integers(2, 7))
        base = rng.integers(0, 4, L, dtype=np.uint8)
        prof = ("ont", "clr", "hifi")[ci % 3]
        seqs = [synth.mutate(base, np.random.default_rng(1000 * ci + j), prof)[0] for j in range(n)]
        seqs = [s for s in seqs if s.size > 0]
        if len(seqs) < 2:
            continue
        buf = C.create_string_buffer(stride * len(seqs))
        for j, s in enumerate(seqs):
            a = ASC[s].tobytes()
            C.memmove(C.addressof(buf) + j * stride + 4, np.uint16(len(a)).tobytes(), 2)
            C.memmove(C.addressof(buf) + j * stride + 6, a + b"\0", len(a) + 1)
        p = lib.poa_to_consensus(C.addressof(buf), len(seqs))
        res = C.string_at(p)
        libc.free(p)
        cases.append((seqs, np.frombuffer(res, dtype=np.uint8)))
    return cases


def make_stage(outdir):
    """Inputs + expected output of the whole seed_cns stage: the reference's own
    lib/nextcorrect.py (run from a temp copy next to oracle/_ref/*.so) at -p 1."""
    import gzip
    import shutil
    import subprocess
    g = synth.make_genome(24000, seed=31, n_repeats=0)
    rs = synth.simulate_reads(g, 28, "ont", seed=32, mu=8.0, sigma=0.35, min_len=1000)
    wd = tempfile.mkdtemp(prefix="ndstage")
    fa = os.path.join(wd, "reads.fa")
    refpipe.write_fasta(fa, [synth.codes_to_ascii(s) for s in rs.seqs])
    idxs, so = refpipe.run_overlap_chain(wd, fa, seed_cutoff=2500)
    lib = os.path.join(wd, "reflib")
    os.makedirs(lib)
    for f in ("nextcorrect.py", "kit.py"):
        shutil.copy(os.path.join("/root/reference/lib", f), lib)
    for f in ("nextcorrect.so", "ovlseq.so"):
        os.symlink(os.path.join(refpipe.REFDIR, f), os.path.join(lib, f))
    os.makedirs(outdir, exist_ok=True)
    for variant, extra in (("default", []), ("split", ["-s"]), ("nobl", ["-b"])):
        out = os.path.join(wd, "cns.%s.fasta" % variant)
        subprocess.run([sys.executable, os.path.join(lib, "nextcorrect.py"), "-f", idxs, "-i", so, "-r", "ont", "-p", "1",
                        "-min_len_seed", "1250", "-o", out] + extra, check=True, cwd=wd, capture_output=True)
        for suffix in ("", ".idx"):
            with open(out + suffix, "rb") as f, gzip.GzipFile(os.path.join(outdir, "cns.%s.fasta%s.gz" % (variant, suffix)),
                                                               "wb", mtime=0) as gz:
                gz.write(f.read())
    db = os.path.join(wd, "db")
    for f in sorted(os.listdir(db)):
        shutil.copy(os.path.join(db, f), outdir)
    shutil.copy(so, os.path.join(outdir, "input.seed.001.sorted.ovl"))
    if os.path.exists(so + ".bl"):
        shutil.copy(so + ".bl", os.path.join(outdir, "input.seed.001.sorted.ovl.bl"))
    return sorted(os.listdir(outdir))


def main():
    assert refpipe.have_ref("nextcorrect.so", "ovlseq.so", "minimap2-nd", "seq_dump", "ovl_sort"), \
        "build the reference first: make -C oracle ref"
    lib = refpipe.ref_cns()
    lib.malloc_vd.argtypes = [C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.c_void_p), C.c_uint64]
    lib.clean_V.argtypes = [C.POINTER(C.c_int), C.c_int]
    lib.destory_vd.argtypes = [C.POINTER(C.c_int), C.c_void_p]
    for f in (lib.align, lib.align_hq):
        f.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(Aln), C.POINTER(C.c_int), C.c_void_p]
        f.restype = None

    r = make_pairs(lib)
    q, qo = ragged(r["q"])
    t, to = ragged(r["t"])
    o, oo = ragged(r["ops"])
    np.savez_compressed(os.path.join(HERE, "align_pairs.npz"), q=q, q_off=qo, t=t, t_off=to, ops=o, ops_off=oo,
                        hq=np.asarray(r["hq"], dtype=np.int32), aln_len=np.asarray(r["aln_len"], dtype=np.int32),
                        t_used=np.asarray(r["t_used"], dtype=np.int32), q_used=np.asarray(r["q_used"], dtype=np.int32))
    print("align_pairs: %d pairs, %d aligned, %d gap-abort, %d failed" % (
        len(r["q"]), sum(1 for n in r["aln_len"] if n > 2), sum(1 for n in r["aln_len"] if n == 2),
        sum(1 for n in r["aln_len"] if n == 0)))

    p = make_piles(lib)
    codes, codes_off = ragged(p["codes"])
    es, eso = ragged(p["exp_seq"])
    np.savez_compressed(os.path.join(HERE, "piles.npz"), codes=codes, codes_off=codes_off,
                        lens=np.asarray(p["lens"], dtype=np.int32), pile_off=np.asarray(p["pile_off"], dtype=np.int64),
                        aln_start=np.asarray(p["aln_start"], dtype=np.uint32),
                        aln_end=np.asarray(p["aln_end"], dtype=np.uint32),
                        max_aln=np.asarray(p["max_aln"], dtype=np.uint32),
                        max_lq=np.asarray(p["max_lq"], dtype=np.uint32),
                        read_type=np.asarray(p["read_type"], dtype=np.int32),
                        fast=np.asarray(p["fast"], dtype=np.int32), split=np.asarray(p["split"], dtype=np.int32),
                        exp_len=np.asarray(p["exp_len"], dtype=np.uint32),
                        exp_ide=np.asarray(p["exp_ide"], dtype=np.float32), exp_seq=es, exp_seq_off=eso)
    print("piles: %d piles, exp_len %s" % (len(p["exp_len"]), p["exp_len"]))

    cases = make_poa(lib)
    flat, off, cnt, res, reso = [], [0], [], [], [0]
    for seqs, r_ in cases:
        cnt.append(len(seqs))
        for s in seqs:
            flat.append(s)
            off.append(off[-1] + s.size)
        res.append(r_)
        reso.append(reso[-1] + r_.size)
    np.savez_compressed(os.path.join(HERE, "poa.npz"), seq=np.concatenate(flat), seq_off=np.asarray(off),
                        count=np.asarray(cnt), res=np.concatenate(res), res_off=np.asarray(reso))
    print("poa: %d cases" % len(cases))
    print("stage:", make_stage(os.path.join(HERE, "stage")))


if __name__ == "__main__":
    main()
