#!/usr/bin/env python
"""bench.py -- corrected bases / s of the MI355X read-correction hot path.

One "step" = one full pass of the correction hot path over a synthetic read set that is resident in HBM:
  (1) overlap stage (`minimap2-nd --step 1` path): minimizer sketch, index, seeds, anchor sort, chain DP, hits of
      the all-vs-all job;
  (2) sort stage (`ovl_sort` path): both directions of every overlap, (seed, match, span) order, coverage-bin
      admission and chimera trimming per seed, `.bl` verdicts -- then the pile admission rules of
      lib/nextcorrect.py:92-143 on the host (vectorised);
  (3) consensus stage (`nextcorrect` path): every pile that came out of (2) -- O(ND) alignments -> MSA -> scoring DP
      -> consensus.
`value` = corrected bases / wall time of (1) + (2) + (3): the whole raw_align -> sort_align -> seed_cns chain of the
reference, with no file in between.  Each stage is checked byte-for-byte against the reference in tests/
(`--no-overlap` / `--analytic-piles` run stage (3) on piles derived from the true read positions instead).

Workload (BASELINE.json configs[1]): synthetic E. coli-sized genome 4.6 Mb, 50x
ONT-profile reads (lognormal, N50 ~ 20-25 kb, sub 3 % / ins 4 % / del 5 %), seed_cutoff 1k (every read is a seed),
`-x ava-ont`, `ovl_sort -k 40` (nextdenovo_amd/synth.py generates the reads).

Launch contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is
started by torch.distributed.run with one rank per GPU.  Piles shard across ranks
with no data-path collective (every rank corrects its own read set: weak scaling);
RCCL is used only for the final corrected-base-count / max-time reduction.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-size", type=float, default=4.6e6)
    ap.add_argument("--depth", type=float, default=50.0)
    ap.add_argument("--profile", default="ont")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="piles in the CPU baseline sample (0 = auto)")
    ap.add_argument("--no-overlap", action="store_true", help="consensus stage only, on analytically derived piles")
    ap.add_argument("--analytic-piles", action="store_true",
                    help="run the overlap stage but feed the consensus stage with piles derived from the true read positions")
    return ap.parse_args()


# ---- CPU baseline leg (the ONLY place bench.py touches oracle/) ---------------------------------
_REF = None


def _ref_worker(item):
    """One fork()ed worker = one `nextcorrect.py -p` worker (lib/nextcorrect.py:183-199):
    calls the compiled reference's nextCorrect() on one pile."""
    global _REF
    import ctypes as C
    seqs, st, en, mal, mlq, rt = item
    if _REF is None:
        class CT(C.Structure):
            _fields_ = [("len", C.c_uint), ("identity", C.c_float), ("seq", C.c_void_p)]
        lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "nextcorrect.so"))
        lib.nextCorrect.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_uint,
                                    C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_float, C.c_uint, C.c_uint,
                                    C.c_int]
        lib.nextCorrect.restype = C.POINTER(CT)
        lib.free_consensus_trimed.argtypes = [C.POINTER(CT)]
        _REF = lib
    lib = _REF
    n = len(seqs)
    cs = (C.c_char_p * n)()
    cs[:] = seqs
    r = lib.nextCorrect(cs, (C.c_uint * n)(*st), (C.c_uint * n)(*en), n, mal, 500, 130, 4, mlq, 0.8, 0, 0, rt)
    ln, ide = r.contents.len, r.contents.identity
    # what the parity block compares: length, the float32 identity bit for bit, md5 of the corrected bases
    # (error seeds -- len 2 / 3 / 4 -- carry no sequence: lib/nextcorrect.c:261-266, 2126)
    digest = hashlib.md5(C.string_at(r.contents.seq, ln)).hexdigest() if ln > 4 else ""
    bits = struct.unpack("<I", struct.pack("<f", ide))[0] if ln > 4 else 0
    lib.free_consensus_trimed(r)
    return ln, bits, digest, ide


def cpu_baseline(rs, piles, read_type, n_sample, n_longest=16):
    """Reference CPU path on a bounded sample of the same workload, all sample piles in
    flight over a fork pool of `cores` workers (the reference's own parallelism model).
    The sample = every k-th pile + the `n_longest` longest seeds (where the device's wide tables, its int32
    guard and its segment stitching matter); the rate is taken over the every-k-th part only, so that it
    stays a representative sample of the workload.  Returns (record, {pile index: (len, identity bits, md5)})."""
    from multiprocessing import get_context
    from nextdenovo_amd import synth
    ref_so = os.path.join(ROOT, "oracle", "_ref", "nextcorrect.so")
    if not os.path.exists(ref_so):
        return None, {}
    cores = max(1, min(os.cpu_count() or 1, 64))
    if n_sample <= 0:
        n_sample = min(len(piles), 4 * cores)
    step = max(1, len(piles) // n_sample)
    idx = list(range(0, len(piles), step))[:n_sample]
    by_len = sorted(range(len(piles)), key=lambda i: -int(piles[i]["recs"][0][3]))
    extra = [i for i in by_len[:n_longest] if i not in set(idx)]

    def item(i):
        seqs, st, en, mal = synth.pile_sequences(rs, piles[i])
        return (seqs, st, en, mal, min(en[0] // 2, 10000), read_type)

    items = [item(i) for i in idx]
    ctx = get_context("fork")
    with ctx.Pool(cores) as pool:
        pool.map(_ref_worker, items[:cores])  # warm: dlopen + page in
        t0 = time.perf_counter()
        got = pool.map(_ref_worker, items, chunksize=1)
        dt = time.perf_counter() - t0
        got_extra = pool.map(_ref_worker, [item(i) for i in extra], chunksize=1) if extra else []
    bases = int(sum(ln for ln, _b, _d, ide in got if ln > 4 and ide >= 0.8))
    ref = {i: g[:3] for i, g in zip(idx + extra, got + got_extra)}
    return {"value": bases / dt, "unit": "corrected bases/s", "cores": cores, "kind": "reference",
            "sample": "%d of %d piles (every %d-th), %d corrected bases in %.2f s wall; compiled reference "
                      "nextcorrect.so via fork pool" % (len(idx), len(piles), step, bases, dt),
            "per_core": bases / dt / cores}, ref


def parity_block(ref, gpu_full, piles):
    """Compare the device's records of the CPU-sample piles (taken from a whole-batch call, the very call shape
    the timed steps make) with the compiled reference's: length, float32 identity bits, md5 of the bases."""
    bad = []
    for i, (ln, bits, digest) in sorted(ref.items()):
        g_ln, g_ide, g_seq = gpu_full[i]
        if ln > 4 or g_ln > 4:
            g_bits = struct.unpack("<I", struct.pack("<f", g_ide))[0] if g_ln > 4 else 0
            g_dig = hashlib.md5(g_seq).hexdigest() if g_ln > 4 else ""
            same = (ln, bits, digest) == (g_ln, g_bits, g_dig)
        else:
            same = ln == g_ln  # error seeds: the convention code only (2 uncorrectable, 3 memory, 4 all clipped)
        if not same:
            bad.append({"pile": i, "seed": int(piles[i]["seed"]), "ref_len": ln, "gpu_len": g_ln})
    lens = [int(piles[i]["recs"][0][3]) + 1 for i in ref]
    return {"piles": len(ref), "mismatch": len(bad), "against": "oracle/_ref/nextcorrect.so (compiled reference)",
            "compared": "len, float32 identity bits, md5(seq)", "longest_seed": max(lens) if lens else 0,
            "seeds_ge_100kb": sum(1 for x in lens if x >= 100000), "error_seeds": sum(1 for v in ref.values() if v[0] <= 4),
            "mismatches": bad[:8]}


def cpu_baseline_overlap(rs_dev, preset):
    """Reference overlapper (oracle/_ref/minimap2-nd --step 1 -t cores) on the same read set: the whole
    all-vs-all job (it finishes in seconds on the host cores), wall time including its index build."""
    import subprocess
    import tempfile
    from nextdenovo_amd import ovl
    exe = os.path.join(ROOT, "oracle", "_ref", "minimap2-nd")
    if not os.path.exists(exe):
        return None
    cores = max(1, min(os.cpu_count() or 1, 64))
    wd = tempfile.mkdtemp(prefix="ndbench")
    p = os.path.join(wd, "reads.2bit")
    ovl.write_2bit(p, rs_dev.ids, rs_dev.lens, rs_dev.words, rs_dev.word_off)
    out = os.path.join(wd, "ref.ovl")
    t0 = time.perf_counter()
    subprocess.run([exe, "--step", "1", "-t", str(cores), "-x", preset, p, p, "-o", out], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    nbytes = os.path.getsize(out)
    bases = int(rs_dev.lens.sum())
    for f in (p, out):
        os.remove(f)
    return {"value": bases / dt, "unit": "query bases/s", "cores": cores, "kind": "reference", "wall_s": dt, "ovl_bytes": nbytes,
            "sample": "whole read set (%d reads, %d bases) all-vs-all, compiled reference minimap2-nd --step 1 -t %d -x %s"
                      % (len(rs_dev), bases, cores, preset)}


def reduce_over_ranks(dist, torch, bases: int, dt: float, device):
    """The path's only collective: sum of corrected bases, max of wall
    time.  RCCL over xGMI on the GPU box ("nccl" backend), gloo in the CPU tests."""
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    b = torch.tensor([bases], dtype=torch.int64, device=device)
    dist.all_reduce(b, op=dist.ReduceOp.SUM)
    return int(b.item()), float(t.item())


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("NDGPU_DEVICE", str(local_rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    from nextdenovo_amd import api, synth

    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    read_type = {"ont": 1, "clr": 2, "hifi": 3}[args.profile]
    analytic = args.analytic_piles or args.no_overlap
    if args.host_threads <= 0:  # the ranks of one node share its cores
        args.host_threads = max(8, (os.cpu_count() or 8) // max(1, world))
    t_gen = time.perf_counter()
    genome = synth.make_genome(int(args.genome_size), seed=42 + 1000 * rank)
    rs = synth.simulate_reads(genome, args.depth, args.profile, seed=43 + 1000 * rank)
    piles = synth.build_piles(rs, seed_cutoff=1000) if analytic else []
    recs, pile_off = synth.flatten_piles(piles) if analytic else (None, None)
    words, word_off, lens = synth.pack_db(rs)
    t_gen = time.perf_counter() - t_gen

    db = api.ReadDB(words, word_off, lens)  # reads resident in HBM from here on

    # overlap + sort stages: the same reads as a .2bit-layout set (read id = index in the DB), all-vs-all; every read is
    # a seed (seed_cutoff 1k <= shortest read), so the run is the single `seed x seed` job of nextDenovo:456-464
    ovl_state = None
    if not args.no_overlap:
        from nextdenovo_amd import overlap
        preset = "ava-ont" if args.profile == "ont" else "ava-pb"
        n_r = len(rs)
        rs_dev = overlap.ReadSet(np.arange(n_r, dtype=np.uint32), lens, words, word_off)
        depth = int(round(args.depth))
        ovl_state = {"opt": overlap.preset(preset), "set": rs_dev, "preset": preset, "stats": None, "bytes": 0, "recs": 0,
                     "wall": 0.0, "sort_wall": 0.0, "asm_wall": 0.0, "sort_stats": None,
                     "k": (depth - 2) if depth <= 30 else min(depth - 5, 40),  # lib/config_parser.py:44
                     "seed_len": lens.astype(np.uint32), "min_seed": int(lens.min()), "last": None}

    def overlap_step():
        """one `minimap2-nd --step 1 seed seed` job (index, map) and, in pipeline mode, `ovl_sort` + the pile assembly
        of lib/nextcorrect.py:92-143 on its records"""
        from nextdenovo_amd import nextcorrect as nc, overlap
        t0 = time.perf_counter()
        with overlap.Index(ovl_state["opt"], ovl_state["set"]) as ix:
            raw = ix.map(ovl_state["set"], ix.mid_occ())
            st = ix.stats()
        ovl_state["recs"] = int(raw.size)
        if analytic:  # stand-alone overlap job: the records are encoded as the .ovl file would be
            ovl_state["bytes"] = len(overlap.encode(raw, np.zeros(2, dtype=np.uint32)))
        t1 = time.perf_counter()
        ovl_state["wall"] += t1 - t0
        if ovl_state["stats"] is None:
            ovl_state["stats"] = st
        else:
            for k, v in st.items():
                ovl_state["stats"][k] += v
        if analytic:
            return None
        srt, bl, sst = overlap.sort_overlaps([raw], ovl_state["seed_len"], ovl_state["min_seed"], ovl_state["k"], 300)
        t2 = time.perf_counter()
        skip = [i for i, kind in bl]  # the .bl blacklist is honoured as lib/nextcorrect.py does by default
        sub, off, seeds = overlap.assemble_piles(srt, ovl_state["seed_len"].size, 500, 500, 130, 10, skip)
        t3 = time.perf_counter()
        ovl_state["sort_wall"] += t2 - t1
        ovl_state["asm_wall"] += t3 - t2
        ovl_state["sort_stats"] = sst
        ovl_state["last"] = (sub, off, seeds, len(bl))
        return sub, off

    def sync():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    cns_wall = [0.0]

    def step():
        r_, o_ = recs, pile_off
        if ovl_state is not None:
            got = overlap_step()
            if got is not None:
                r_, o_ = got
        t_c = time.perf_counter()
        res = db.correct_piles(r_, o_, read_type=read_type, host_threads=args.host_threads, lengths_only=True)
        cns_wall[0] += time.perf_counter() - t_c
        # accepted records exactly as lib/nextcorrect.py:236 (len >= min_len_seed(=seed_cutoff/2), identity >= ratio)
        return sum(ln for ln, ide in res if ln >= 500 and ln > 4 and ide >= 0.8)

    for _ in range(args.warmup):
        step()
    cns_wall[0] = 0.0
    api.reset_stats()
    if ovl_state is not None:
        ovl_state["stats"], ovl_state["wall"], ovl_state["sort_wall"], ovl_state["asm_wall"] = None, 0.0, 0.0, 0.0
    sync()
    t0 = time.perf_counter()
    bases = 0
    for _ in range(args.steps):
        bases += step()
    sync()
    dt = time.perf_counter() - t0
    st = api.stats()

    total_bases = bases
    max_dt = dt
    if dist is not None:
        total_bases, max_dt = reduce_over_ranks(dist, torch, bases, dt, "cuda")

    if rank == 0:
        if not analytic:  # the piles the last step really corrected (for the config line and the CPU sample)
            sub, off, seeds, n_bl = ovl_state["last"]
            piles = [{"seed": int(seeds[i]), "recs": sub[int(off[i]):int(off[i + 1])]} for i in range(seeds.size)]
            recs = sub
        # Roofline of the dominant kernel by GPU time: K10 score_fast (scoring DP).  Algorithmic
        # bytes per launch = every MSA cell table entry read once (start,len: 8 B) and its best_pp /
        # best_link written once (8 B) + every link read once (pp, ppp, count: 12 B) + the per-column
        # metadata (5 x 4 B).  K7 (O(ND) forward) is reported alongside: 2-bit operands read once +
        # 1 trace bit per evaluated cell + 4 B min_k per edit step.
        k10_launches = max(1, st["score_launches"])
        k10_bytes = (16.0 * st["cells_msa"] + 12.0 * st["links"] + 20.0 * st["path_items"]) / k10_launches
        k10_ms = st["score_ms"] / k10_launches
        achieved = k10_bytes / (k10_ms * 1e-3) / 1e9 if k10_ms > 0 else 0.0
        # HBM bytes per K10 launch from the PMC passes committed under profiles/ (bench.py cannot run rocprofv3 on itself):
        # used only when this run is the workload those passes profiled (seeded data: the launches are the same ones)
        traffic, traffic_note = None, None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_k10_traffic.json")) as f:
                pm = json.load(f)
            w_ = pm["workload"]
            if (not analytic and ovl_state is not None and int(args.genome_size) == w_["genome_size"] and int(args.depth) == w_["depth"]
                    and args.profile == w_["profile"] and abs(k10_launches / args.steps - pm["launches_per_step"]) < 0.5):
                traffic = (pm["fetch_kb_per_launch"] + pm["write_kb_per_launch"]) * 1024.0
                traffic_note = pm["source"]
        except (OSError, KeyError, ValueError):
            pass
        launches = max(1, st["forward_launches"])
        alg_bytes = (st["seq_bases"] / 4.0 + st["cells"] / 8.0 + 4.0 * st["d_steps"]) / launches
        avg_ms = st["forward_ms"] / launches
        k7_achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        peak = 8000.0
        out = {
            "metric": "corrected bases/sec",
            "value": total_bases / max_dt,
            "unit": "bases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": max_dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": "synthetic E. coli-like %.1f Mb, %gx %s reads (lognormal mu 9.55 sigma 0.75), "
                                   "seed_cutoff 1k, 1 read set per GPU" % (args.genome_size / 1e6, args.depth,
                                                                          args.profile),
                       "reads_per_gpu": len(rs), "read_bases_per_gpu": rs.total_bases(), "piles_per_gpu": len(piles),
                       "overlaps_per_gpu": int(recs.shape[0]), "sharding": "piles, weak (one read set per rank)",
                       "piles_from": "analytic (true read positions)" if analytic else
                       "the step's own overlap -> ovl_sort -> pile assembly chain on the device",
                       "datagen_s": round(t_gen, 1)},
            "roofline": {"bound": "hbm", "kernel": "score_fast_kernel (K10 scoring DP)", "achieved": achieved,
                         "peak": peak, "unit": "GB/s", "frac": achieved / peak, "traffic": traffic,
                         "traffic_unit": "bytes per launch (FETCH_SIZE + WRITE_SIZE)", "traffic_source": traffic_note,
                         "traffic_over_algorithmic": (traffic / k10_bytes) if traffic and k10_bytes else None,
                         "alg_bytes_per_launch": k10_bytes, "avg_launch_ms": k10_ms, "launches": int(k10_launches),
                         "note": "latency-bound dependent chain (one wave per seed), not bandwidth-bound",
                         "k7_ond_forward": {"achieved": k7_achieved, "frac": k7_achieved / peak,
                                            "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_ms,
                                            "launches": int(launches),
                                            "cells_per_s": st["cells"] / (st["forward_ms"] * 1e-3)
                                            if st["forward_ms"] > 0 else 0.0}},
            "counters": {k: st[k] for k in ("tasks", "wide_tasks", "cells", "d_steps", "max_band", "piles", "tags",
                                            "cells_msa", "links", "path_items")},
            "kernel_ms": {k: round(st[k], 2) for k in ("forward_ms", "traceback_ms", "tags_ms", "links_ms", "score_ms",
                                                       "backtrack_ms", "extract_ms")},
            "consensus_ms_per_step": cns_wall[0] / args.steps * 1e3,
        }
        if ovl_state is not None:
            ost = ovl_state["stats"]
            q_bases = int(ovl_state["set"].lens.sum())
            o_ms = ovl_state["wall"] / args.steps * 1e3
            # dominant overlap kernels by time; algorithmic bytes of the anchor pipeline:
            # 16 B per anchor written by K3, sorted (LSD passes of 8 bits: 2 x 16 B per pass), read once by K4
            passes = 6
            a_bytes = ost["anchors"] / args.steps * (16.0 + passes * 32.0 + 16.0)
            gpu_ms = sum(ost[k] for k in ("sketch_ms", "index_sort_ms", "seed_ms", "sort_ms", "exact_sort_ms", "chain_ms", "hits_ms")) / args.steps
            out["overlap"] = {
                "included_in_value": True,
                "value": q_bases * world / (o_ms * 1e-3), "unit": "query bases/s (all-vs-all, index build included)",
                "ms_per_step": o_ms, "gpu_kernel_ms_per_step": gpu_ms,
                "kernel_ms": {k: round(ost[k] / args.steps, 3) for k in ("sketch_ms", "index_sort_ms", "seed_ms", "sort_ms", "exact_sort_ms",
                                                                        "chain_ms", "hits_ms")},
                "counters": {k: int(ost[k] // args.steps) for k in ("bases_sketched", "minimizers", "anchors", "tie_reads", "chain_cells",
                                                                    "chains", "overlaps")},
                "ovl_bytes": ovl_state["bytes"], "records": ovl_state["recs"],
                "sketch_gsymbols_per_s": ost["bases_sketched"] / (ost["sketch_ms"] * 1e-3) / 1e9 if ost["sketch_ms"] > 0 else 0.0,
                "anchor_pipeline_alg_GBps": a_bytes / (gpu_ms * 1e-3) / 1e9 if gpu_ms > 0 else 0.0,
                "chain_gcells_per_s": ost["chain_cells"] / (ost["chain_ms"] * 1e-3) / 1e9 if ost["chain_ms"] > 0 else 0.0,
            }
            if not analytic:
                out["overlap"]["sort"] = {"ms_per_step": ovl_state["sort_wall"] / args.steps * 1e3,
                                          "gpu_ms": ovl_state["sort_stats"]["gpu_ms"],
                                          "candidates": int(ovl_state["sort_stats"]["candidates"]),
                                          "kept": int(ovl_state["sort_stats"]["kept"]), "blacklisted": int(ovl_state["last"][3]),
                                          "k": ovl_state["k"]}
                out["overlap"]["pile_assembly_ms_per_step"] = ovl_state["asm_wall"] / args.steps * 1e3
        parity_fail = False
        if not args.no_cpu_baseline and world == 1:  # the CPU leg runs at N = 1 only
            out["cpu_baseline"], ref = cpu_baseline(rs, piles, read_type, args.cpu_sample)
            if ref:  # one more (untimed) whole-batch call that keeps the sequences: what the timed steps computed
                full = db.correct_piles(recs, off if not analytic else pile_off, read_type=read_type,
                                        host_threads=args.host_threads)
                out["parity"] = parity_block(ref, full, piles)
                parity_fail = out["parity"]["mismatch"] != 0
            if ovl_state is not None:
                out["overlap"]["cpu_baseline"] = cpu_baseline_overlap(ovl_state["set"], ovl_state["preset"])
        print(json.dumps(out))
        if parity_fail:
            db.close()
            sys.exit("bench.py: the device's records differ from the reference's (see \"parity\")")
    db.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
