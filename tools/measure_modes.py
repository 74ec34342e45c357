#!/usr/bin/env python
"""Timings of the overlap-engine paths beyond the default `--step 1`: `--step 1 --mode 3` (HiFi end extension), `--step 2 --mode 0`,
`--step 2` (the re-alignment) and the ksw2-extd2 batch kernel, each next to the compiled reference on the same box's host cores
(minimap2-nd -t <cores>; ksw_extd2_sse single thread).  Run on the GPU box:
    python tools/measure_modes.py gpurun_out/<tag>/measure_modes.json
"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from nextdenovo_amd import minimap2_nd, overlap, synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
CORES = os.cpu_count() or 1


def write_fasta(path, seqs, first_id=1):
    with open(path, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">%d %d 0.99\n%s\n" % (first_id + i, s.size, synth.codes_to_ascii(s).decode()))


def timed(fn, repeat=2):
    best = None
    for _ in range(repeat):
        t0 = time.perf_counter()
        r = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best, r


def run_ref(argv):
    t0 = time.perf_counter()
    subprocess.run([os.path.join(REF, "minimap2-nd"), *argv], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return time.perf_counter() - t0


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "measure_modes.json"
    only = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else {"modes", "c", "ksw"}   # which blocks to run
    small = bool(os.environ.get("ND_MEASURE_SMALL"))   # (a dry run of the script itself on tiny inputs)
    wd = tempfile.mkdtemp(prefix="ndmodes")
    res = {"host_cores": CORES}
    g = synth.make_genome(40000 if small else 3000000, seed=91, n_repeats=5)
    rs = synth.simulate_reads(g, 30, "hifi", seed=92, mu=9.3, sigma=0.35, min_len=3000)
    fa = os.path.join(wd, "cns.fasta")
    write_fasta(fa, rs.seqs)
    bases = int(sum(s.size for s in rs.seqs))
    res["reads"] = {"n": len(rs.seqs), "bases": bases, "what": "3 Mb genome, 30x corrected-read-like (HiFi error profile) reads, lognormal mu 9.3"}
    have_ref = os.path.exists(os.path.join(REF, "minimap2-nd"))
    cases = [
        ("step1_mode3_ava_hifi", ["--step", "1", "--mode", "3", "-x", "ava-hifi"]),
        ("step2_mode0", ["--step", "2", "--mode", "0", "--dual=yes", "-x", "ava-ont", "-k", "17", "-w", "17", "--minlen", "2000"]),
        ("step2_default_mode2", ["--step", "2", "--dual=yes", "-x", "ava-ont", "-k", "17", "-w", "17", "--minlen", "2000"]),
    ]
    for name, argv in (cases if "modes" in only else []):
        out = os.path.join(wd, name + ".ovl")
        minimap2_nd.run([*argv, "-t", "8", fa, fa, "-o", out])   # warm-up (HIP initialisation, allocator)
        dev_s, _ = timed(lambda: minimap2_nd.run([*argv, "-t", "8", fa, fa, "-o", out]))
        entry = {"device_s": dev_s, "device_query_bases_per_s": bases / dev_s, "ovl_bytes": os.path.getsize(out),
                 "includes": "FASTA parsing, packing, index build, mapping, host filters, encoding, file output"}
        if have_ref:
            ref_out = os.path.join(wd, name + ".ref.ovl")
            ref_s = run_ref([*argv, "-t", str(CORES), fa, fa, "-o", ref_out])
            entry["reference_s"] = ref_s
            entry["reference_threads"] = CORES
            entry["identical"] = open(out, "rb").read() == open(ref_out, "rb").read()
            entry["speedup"] = ref_s / dev_s
        res[name] = entry
        print(name, json.dumps(entry), flush=True)

    # --step 1 -c on raw reads (ONT error profile): base-level alignment through every chain (csrc/ovl_cigar.cpp)
    if "c" in only:
        measure_c(res, wd, have_ref, small)
    if "ksw" in only:
        measure_ksw(res, small)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)


def measure_c(res, wd, have_ref, small):
    g2 = synth.make_genome(30000 if small else 1000000, seed=93, n_repeats=5)
    rs2 = synth.simulate_reads(g2, 20, "ont", seed=94, mu=9.3, sigma=0.5, min_len=2000)
    fa2 = os.path.join(wd, "raw.fasta")
    write_fasta(fa2, rs2.seqs)
    bases2 = int(sum(s.size for s in rs2.seqs))
    argv = ["--step", "1", "-c", "--dual=yes", "-x", "ava-ont"]
    out = os.path.join(wd, "step1_c.ovl")
    minimap2_nd.run([*argv, "-t", "8", fa2, fa2, "-o", out])
    dev_s, _ = timed(lambda: minimap2_nd.run([*argv, "-t", "8", fa2, fa2, "-o", out]))
    entry = {"reads": {"n": len(rs2.seqs), "bases": bases2, "what": "1 Mb genome, 20x ONT-profile raw reads"}, "device_s": dev_s,
             "device_query_bases_per_s": bases2 / dev_s, "ovl_bytes": os.path.getsize(out)}
    if have_ref:
        ref_out = os.path.join(wd, "step1_c.ref.ovl")
        ref_s = run_ref([*argv, "-t", str(CORES), fa2, fa2, "-o", ref_out])
        entry.update(reference_s=ref_s, reference_threads=CORES, identical=open(out, "rb").read() == open(ref_out, "rb").read(), speedup=ref_s / dev_s)
    # where the time goes: the same job through the library's entry point (index + ndgpu_ovl_map_cigar), with the call's own clock
    words, word_off, lens = synth.pack_db(rs2)
    dset = overlap.ReadSet(np.arange(1, len(rs2.seqs) + 1, dtype=np.uint32), lens, words, word_off)
    opt = minimap2_nd.build_opt(minimap2_nd.parse_argv([*argv, "a", "b"]))
    with overlap.Index(opt, dset) as ix:
        mid = ix.mid_occ()
        ix.map_cigar(dset, dset, mid, want_stats=True)   # (warm)
        t0 = time.perf_counter()
        recs, st = ix.map_cigar(dset, dset, mid, want_stats=True)
        call_s = time.perf_counter() - t0
    host_ns = st["total_ns"] - st["chains_ns"] - st["ksw_ns"] - st["ksw_ll_ns"]
    entry["split"] = {"map_cigar_call_s": call_s, "records": int(recs.shape[0]),
                      "chains_on_device_s": st["chains_ns"] * 1e-9, "ksw_extd2_batches_s": st["ksw_ns"] * 1e-9, "ksw_ll_batches_s": st["ksw_ll_ns"] * 1e-9,
                      "host_chain_walk_cigar_join_filters_s": host_ns * 1e-9, "dp_cells": st["cells"], "first_pass": st["first_pass"],
                      "second_pass": st["second_pass"], "inversion_tests": st["inversion_tests"], "chains": st["chains"],
                      "dp_gcells_per_s_in_batches": st["cells"] / max(1, st["ksw_ns"]),
                      "note": "ksw batches = upload + kernel + backtrack + download of ndgpu_ksw_extd2_batch; host = chain walking, "
                              "CIGAR joining, z-drop bookkeeping, filters, sort (host threads)"}
    res["step1_c_ava_ont"] = entry
    print("step1_c_ava_ont", json.dumps(entry), flush=True)


def measure_ksw(res, small):
    # ksw2-extd2: a batch of gap-filling / extension problems as mm_align1 would hand them over
    import ksw_util as K
    rng = np.random.default_rng(5)
    lib = overlap.load()
    probs = []
    for n in range(60 if small else 6000):
        L = int(rng.choice([200, 500, 1000, 2000, 4000]))
        t = rng.integers(0, 4, L).astype(np.uint8)
        q = t.copy()
        for k in range(L // 12):
            i = int(rng.integers(0, q.size - 3))
            r = rng.random()
            if r < .5:
                q[i] = (q[i] + 1) % 4
            elif r < .75:
                q = np.delete(q, slice(i, i + int(rng.integers(1, 3))))
            else:
                q = np.insert(q, i, rng.integers(0, 4, int(rng.integers(1, 3))))
        probs.append(dict(q=q.astype(np.uint8), t=t, mat=K.matrix(2, 4, 1), gaps=(4, 2, 24, 1), w=751, zdrop=400, end_bonus=5, flag=K.F_EXTZ_ONLY))
    cells = float(sum(p["q"].size * min(p["t"].size, 2 * 751 + 1) for p in probs))
    K.call_batch(lib, probs[:64])
    dev_s, _ = timed(lambda: K.call_batch(lib, probs))
    entry = {"problems": len(probs), "band_cells": cells, "device_s": dev_s, "device_gcells_per_s": cells / dev_s / 1e9,
             "includes": "host packing, H2D, kernel, backtrack, D2H, CIGAR copies (ndgpu_ksw_extd2_batch)"}
    refso = os.path.join(REF, "libksw2ref.so")
    if os.path.exists(refso):
        ref = C.CDLL(refso)
        sample = probs[::20]
        t0 = time.perf_counter()
        for p in sample:
            K.call_sse(ref, p["q"], p["t"], p["mat"], *p["gaps"], p["w"], p["zdrop"], p["end_bonus"], p["flag"])
        ref_s = time.perf_counter() - t0
        sc = float(sum(p["q"].size * min(p["t"].size, 2 * 751 + 1) for p in sample))
        entry["reference_gcells_per_s_one_core"] = sc / ref_s / 1e9
        entry["reference_sample"] = "%d of the problems, ksw_extd2_sse (SSE4.1) through ctypes, one thread" % len(sample)
    res["ksw2_extd2_batch"] = entry
    print("ksw2", json.dumps(entry), flush=True)


if __name__ == "__main__":
    main()
