#!/usr/bin/env python
"""The overlap stage with options drawn at random -- preset, k, w, -n, -m, -f INT[,INT], --dual, --mode 3, the anchor budget of a batch --
on small seeded read sets with repeats: the device library (under the kernel interpreter when there is no GPU: tests/simt) against the
oracle's `.ovl` bytes.  usage: fuzz_overlap_options.py <seed> <cases> [gpu]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "simt"))
import mm_util as M  # noqa: E402


def main():
    seed, cases = int(sys.argv[1]), int(sys.argv[2])
    on_gpu = len(sys.argv) > 3 and sys.argv[3] == "gpu"
    from nextdenovo_amd import overlap, synth
    if not on_gpu:
        import build_simt
        os.environ.setdefault("NDGPU_CONTEXTS", "1")
        overlap._lib = overlap._bind(C.CDLL(build_simt.build_overlap()))
    olib = M.bind(C.CDLL(os.path.join(ROOT, "oracle", "libndoracle.so")))
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(cases):
        profile, preset = [("ont", "ava-ont"), ("clr", "ava-pb"), ("hifi", "ava-hifi")][int(rng.integers(0, 3))]
        g = synth.make_genome(int(rng.integers(15000, 40000)), seed=int(rng.integers(1 << 30)), n_repeats=int(rng.integers(0, 5)), repeat_len=int(rng.integers(300, 1500)))
        if rng.random() < 0.5:   # a tandem array: many equal minimizers, occurrence thresholds matter
            unit = rng.integers(0, 4, int(rng.integers(5, 80))).astype(np.uint8)
            blk = np.tile(unit, int(rng.integers(10, 60)))
            at = int(rng.integers(0, g.size - blk.size))
            g[at:at + blk.size] = blk
        rs = synth.simulate_reads(g, float(rng.integers(6, 16)), profile, seed=int(rng.integers(1 << 30)), mu=7.6, sigma=0.4, min_len=600)
        n = len(rs.seqs)
        ids = np.arange(n, dtype=np.uint32) + np.uint32(rng.integers(0, 1000))
        lens = np.asarray([s.size for s in rs.seqs], dtype=np.uint32)
        words = [synth.pack_2bit_msb(s) for s in rs.seqs]
        woff = np.zeros(n, dtype=np.uint64)
        woff[1:] = np.cumsum([w.size for w in words])[:-1]
        dset = overlap.ReadSet(ids, lens, np.concatenate(words), woff)
        off = np.zeros(n, dtype=np.uint64)
        off[1:] = np.cumsum(lens.astype(np.uint64))[:-1]
        oset = (ids, lens, np.concatenate(rs.seqs).astype(np.uint8), off)
        dual = bool(rng.integers(0, 2))
        kw = {}
        if rng.random() < 0.6:
            kw["k"] = int(rng.choice([9, 12, 15, 19, 24, 28, 29, 30, 31, 33, 40, 51, 66]))
            kw["w"] = int(rng.choice([1, 3, 5, 10, 19, 40]))
        if rng.random() < 0.5:
            kw["min_cnt"] = int(rng.choice([1, 2, 3, 5]))
            kw["min_sc"] = int(rng.choice([15, 40, 100]))
        if rng.random() < 0.3:
            kw["minlen"] = int(rng.choice([14, 200, 500]))
        mid, max_occ = 0, 0
        if rng.random() < 0.6:
            mid = int(rng.choice([2, 3, 5, 8, 20, 50]))
            if rng.random() < 0.7:
                max_occ = int(mid * rng.choice([2, 5, 40]))
        mode3 = rng.random() < 0.25
        budget = int(rng.choice([3000, 20000, 100000000]))
        os.environ["NDGPU_OVL_BATCH_ANCHORS"] = str(budget)
        oo = M.preset(preset, dual, max_occ=max_occ, **kw)
        frac = 1e-4 if preset == "ava-hifi" else 2e-4
        want, mid_used = M.step1(olib, oo, oset, oset, mid_occ_frac=frac, mid_occ=mid, mode3=mode3)
        o = overlap.preset(preset)
        o.no_dual = 0 if dual else 1
        o.max_occ = max_occ
        for name, v in kw.items():
            setattr(o, "min_chain_score" if name == "min_sc" else name, v)
        if mode3:
            o.mode = 3
        try:
            with overlap.Index(o, dset) as ix:
                m = mid if mid > 0 else ix.mid_occ()
                recs = ix.map(dset, m)
                st = ix.stats()
            got = overlap.encode(recs, np.zeros(2, dtype=np.uint32))
            ok = got == want and m == mid_used
        except Exception as e:   # noqa: BLE001
            ok, got, st = False, b"", {"error": repr(e)}
        if not ok:
            bad += 1
        print("case %d: %s %s dual %d %s mid %d max_occ %d mode3 %d budget %d reads %d: %d bytes, batches %s rechained %s -> %s"
              % (it, profile, preset, dual, kw, mid, max_occ, mode3, budget, n, len(want), st.get("batches"), st.get("rechained"), "ok" if ok else "MISMATCH %d" % len(got)), flush=True)
    print("seed %d: %d cases, %d bad" % (seed, cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
