#!/usr/bin/env python
"""Random read sets (insertions, inversions, deletions spliced into a third of the reads) and random options against `--step 1 -c`:
    python tools/fuzz_cigar.py oracle SEED N    the oracle (oracle/cigar_oracle.c) against the compiled reference binary (needs oracle/_ref)
    python tools/fuzz_cigar.py device SEED N    the device path (ndgpu_ovl_map_cigar) against the oracle (needs a GPU; NDGPU_SIMT=1 binds the
                                                library to the interpreted build instead: minutes per case)
Round 3: 69 `oracle` cases (one default found wrong and fixed: this minimap2 has no --cap-sw-mem default), 8 `device` cases under the
interpreter, all equal."""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mm_util as M  # noqa: E402
from nextdenovo_amd import overlap, synth  # noqa: E402


def reads(rng, g_lo, g_hi, d_lo, d_hi):
    g = synth.make_genome(int(rng.integers(g_lo, g_hi)), seed=int(rng.integers(1, 10 ** 6)), n_repeats=int(rng.integers(0, 5)),
                          repeat_len=int(rng.integers(400, 2500)))
    rs = synth.simulate_reads(g, float(rng.uniform(d_lo, d_hi)), "ont", seed=int(rng.integers(1, 10 ** 6)), mu=float(rng.uniform(8.2, 9.1)),
                              sigma=float(rng.uniform(0.3, 0.6)), min_len=1500)
    out = []
    for s in rs.seqs:
        s, r = s.copy(), rng.random()
        if r < 0.25 and s.size > 3500:
            p = int(s.size * rng.uniform(0.2, 0.8))
            s = np.concatenate([s[:p], rng.integers(0, 4, int(rng.integers(100, 1500))).astype(np.uint8), s[p:]])
        elif r < 0.5 and s.size > 3500:
            p, ln = int(s.size * rng.uniform(0.2, 0.7)), int(rng.integers(300, 1500))
            if p + ln < s.size - 300:
                s[p:p + ln] = synth.revcomp_codes(s[p:p + ln])
        elif r < 0.6 and s.size > 3500:
            p, ln = int(s.size * rng.uniform(0.3, 0.6)), int(rng.integers(200, 1200))
            s = np.concatenate([s[:p], s[p + ln:]])
        out.append(s)
    return out


def options(rng):
    """-> (argv of the reference, oracle MMOpt keywords, scoring keywords)"""
    extra, kw, ao = ["-c"], {}, {}
    if rng.random() < 0.4:
        z = int(rng.choice([100, 200, 600]))
        zi = int(rng.choice([50, 100, z]))
        extra += ["-z", "%d,%d" % (z, zi)]
        ao.update(zdrop=z, zdrop_inv=zi)
    if rng.random() < 0.3:
        s_ = int(rng.choice([40, 120, 200]))
        extra += ["-s", str(s_)]
        ao["min_dp_max"] = s_
    if rng.random() < 0.3:
        bw = int(rng.choice([100, 300, 1000]))
        extra += ["-r", str(bw)]
        kw["bw"] = bw
    if rng.random() < 0.3:
        a, b = [(1, 2), (3, 5), (2, 6)][int(rng.integers(0, 3))]
        extra += ["-A", str(a), "-B", str(b)]
        ao.update(a=a, b=b)
    if rng.random() < 0.3:
        o = [(4, 24), (6, 30), (2, 12)][int(rng.integers(0, 3))]
        extra += ["-O", "%d,%d" % o]
        ao.update(q=o[0], q2=o[1])
    if rng.random() < 0.2:
        extra += ["--dvt"]
        kw["dvt"] = 1
    if rng.random() < 0.2:
        m = int(rng.choice([40, 60, 200]))
        extra += ["-m", str(m)]
        kw["min_sc"] = m
    if rng.random() < 0.2:
        gg = int(rng.choice([2000, 5000]))
        extra += ["-g", str(gg)]
        kw["max_gap"] = gg
    if rng.random() < 0.2:
        nn = int(rng.choice([2, 4, 6]))
        extra += ["-n", str(nn)]
        kw["min_cnt"] = nn
    return extra, kw, ao


def main():
    mode, seed, n_cases = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rng = np.random.default_rng(seed)
    olib = M.bind(C.CDLL(os.path.join(ROOT, "oracle", "libndoracle.so")))
    if mode == "device" and os.environ.get("NDGPU_SIMT"):
        sys.path.insert(0, os.path.join(ROOT, "tests", "simt"))
        import build_simt
        os.environ.setdefault("NDGPU_CONTEXTS", "1")
        overlap._lib = overlap._bind(C.CDLL(build_simt.build_overlap()))
    bad = 0
    for it in range(n_cases):
        seqs = reads(rng, 15000, 45000, 8, 22) if mode == "oracle" else reads(rng, 12000, 22000, 8, 14)
        preset, (extra, kw, ao_kw) = str(rng.choice(["ava-ont", "ava-pb"])), options(rng)
        mao = M.aln_opt(**ao_kw)
        t0 = time.time()
        if mode == "oracle":
            dual = bool(rng.random() < 0.5)
            wd = tempfile.mkdtemp(prefix="fz")
            seed_f, part_f = M.dump_reads(wd, [synth.codes_to_ascii(s) for s in seqs], seed_cutoff=int(rng.choice([3000, 5000])))
            t, q = seed_f, (part_f if dual and part_f else seed_f)
            try:
                want = M.ref_step1(t, q, os.path.join(wd, "ref.ovl"), preset, dual, tuple(extra), threads=4)
            except subprocess.CalledProcessError:
                print(it, "reference failed", preset, extra, flush=True)
                continue
            got, _ = M.step1_cigar(olib, M.preset(preset, dual, **kw), mao, M.load_set(t), M.load_set(q))
            where = wd
        else:
            n = len(seqs)
            ids, lens = np.arange(1, n + 1, dtype=np.uint32), np.asarray([s.size for s in seqs], dtype=np.uint32)
            words = [synth.pack_2bit_msb(s) for s in seqs]
            woff = np.zeros(n, dtype=np.uint64)
            woff[1:] = np.cumsum([w.size for w in words])[:-1]
            rset = overlap.ReadSet(ids, lens, np.concatenate(words), woff)
            off = np.zeros(n, dtype=np.uint64)
            off[1:] = np.cumsum(lens.astype(np.uint64))[:-1]
            oset = (ids, lens, np.concatenate(seqs).astype(np.uint8), off)
            o, ao = overlap.preset(preset), overlap.aln_opt(**ao_kw)
            for k, v in kw.items():
                setattr(o, {"min_sc": "min_chain_score"}.get(k, k), v)
            with overlap.Index(o, rset) as ix:
                mid = ix.mid_occ()
                recs = ix.map_cigar(rset, rset, mid, ao)
            got = overlap.encode(recs, np.zeros(2, dtype=np.uint32))
            want, _ = M.step1_cigar(olib, M.preset(preset, False, **kw), mao, oset, oset, mid_occ=mid)
            where = ""
        ok = got == want
        bad += not ok
        print(it, "equal" if ok else "DIFFER", preset, extra, len(got), len(want), "%.1fs" % (time.time() - t0), "" if ok else where, flush=True)
    print("mismatches", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
