#!/usr/bin/env python
"""bench.py -- corrected bases / s of the MI355X read-correction hot path.

One "step" = one full pass of the correction stage over a synthetic read set whose 2-bit reads are resident in HBM:
  (1) overlap stage (`minimap2-nd --step 1` path): minimizer sketch, index, seeds, anchor sort, chain DP, hits of every
      raw_align job of the rank's seed file (nextDenovo:426-467);
  (2) sort stage (`ovl_sort` path): both directions of every overlap, (seed, match, span) order, coverage-bin admission and
      chimera trimming per seed, `.bl` verdicts -- then the pile admission rules of lib/nextcorrect.py:92-143;
  (3) consensus stage (`nextcorrect` path): every pile that came out of (2) -- O(ND) alignments -> MSA -> scoring DP ->
      consensus.
`value` = corrected bases / wall time of (1) + (2) + (3): the whole raw_align -> sort_align -> seed_cns chain of the reference
with no file in between.  Each stage is checked byte-for-byte against the reference in tests/, and the run itself compares the
records of its CPU-sample piles with the compiled reference (`parity`).

Workloads = BASELINE.json configs, `--config N`:
  2 (default)  E. coli-like 4.6 Mb, 50x ONT, N50 ~ 20 kb, seed_cutoff 1k, -x ava-ont, ovl_sort -k 40
  3            D. melanogaster-like 140 Mb with 20 % interspersed repeats, 40x ONT (large pile depth)
  4            A. thaliana-like 120 Mb, 60x PacBio CLR, -x ava-pb
  5            human chr1-like 250 Mb with 45 % repeats, 30x ultra-long ONT (N50 ~ 100 kb, <= 1 Mb)
(nextdenovo_amd/synth.py generates the reads; configs 3-5 on all host cores.)

Sharding = the reference's own (nextdenovo_amd/stage.py): seeds are dealt round robin into `seed_cutfiles` seed files
(util/seq_dump.c:87-92) and a seed file is corrected from the overlaps between its seeds and every read.  `--gpus N` makes
N seed files of ONE read set and gives seed file r to rank r (strong scaling: the total work is fixed); no rank needs
anything another rank computed, RCCL carries only the final {corrected bases, seeds, wall time} reduction.
`--seed-files M` on one GPU corrects seed file `--shard` of M: the single-GPU share of an M-GPU run (how configs 4 and 5,
which BASELINE.json places on 8 GPUs, are run on the one MI355X at hand).

Launch contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 started by torch.distributed.run, one rank per GPU.
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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 8 after 2 warm-up steps for config 2 -- the line of two calls in flight needs a few steps to "
                         "show its rate; 2 after 1 for the genome-scale configs, whose steps take seconds)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json configs[N-1]")
    ap.add_argument("--seed-files", type=int, default=0, help="seed_cutfiles (default: the number of GPUs)")
    ap.add_argument("--shard", type=int, default=0, help="with one GPU and --seed-files M: which seed file to correct")
    ap.add_argument("--genome-size", type=float, default=0, help="override: uniform random genome of this size")
    ap.add_argument("--depth", type=float, default=0)
    ap.add_argument("--profile", default="")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap-baseline", action="store_true",
                    help="skip the reference overlapper / ovl_sort leg of the CPU baseline (minutes on a genome-scale read set); the consensus leg and the parity block stay")
    ap.add_argument("--lengths-only", action="store_true",
                    help="the timed steps skip the hand-over of the corrected sequences and the cns.fasta / .idx write (A/B runs only: "
                         "the output is part of the stage)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="piles in the CPU baseline sample (0 = auto)")
    ap.add_argument("--no-overlap", action="store_true", help="consensus stage only, on analytically derived piles")
    ap.add_argument("--consensus-depth", type=int, default=2, help="consensus calls in flight (1: a step's consensus ends before the next begins)")
    ap.add_argument("--producers", type=int, default=0,
                    help="threads (a Shard each) that compute the piles of later steps side by side (default: 1 on one GPU -- measured on config 2, a "
                         "second one takes the line's waits for piles from 160 to 70 ms per step and leaves the step where it was: the device is the "
                         "bound --, 2 per rank of several when the rank has six CPUs or more: a rank's step is bound by its stage)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not start the overlap / sort / pile-admission stage of the next step while the consensus of this one runs")
    ap.add_argument("--no-exchange", action="store_true",
                    help="N > 1: every rank computes the mirror jobs it needs itself instead of taking them from the rank that owns them "
                         "(nextdenovo_amd/stage.py: Exchange)")
    ap.add_argument("--analytic-piles", action="store_true",
                    help="run the overlap stage but feed the consensus stage with piles derived from the true read positions")
    args = ap.parse_args(argv)
    small = args.config == 2
    if args.steps is None:
        args.steps = 8 if small else 2
    if args.warmup is None:
        args.warmup = 2 if small else 1
    return args


# ---- roofline bookkeeping -----------------------------------------------------------------------
# Algorithmic bytes of the consensus kernels, from the run's own counters (operands once as 2-bit words,
# one trace bit per evaluated (d, k) cell, outputs once).  Names are the kernels' names in a rocprofv3 kernel trace.
def kernel_models(st):
    m = {}
    m["ond_forward_kernel"] = {
        "label": "K7 ond_forward (banded O(ND) forward sweep, one wavefront per alignment)",
        "alg_bytes": st["seq_bases"] / 4.0 + st["cells"] / 8.0,
        "formula": "seq_bases / 4 (2-bit operands, read once) + cells / 8 (one trace bit per evaluated cell, written once)",
        "ms": st["forward_ms"], "launches": st["forward_launches"]}
    m["ond_traceback_kernel"] = {
        "label": "K8a ond_traceback (one lane per alignment)",
        "alg_bytes": st["seq_bases"] / 4.0 + st["d_steps"] / 8.0 + st["columns"] / 4.0,
        "formula": "seq_bases / 4 (operands) + d_steps / 8 (the path's move bit of every edit step) + columns / 4 (2-bit column kinds out)",
        "ms": st["traceback_ms"], "launches": st["traceback_launches"]}
    m["tb_walk_kernel"] = {
        "label": "K8a in segments: tb_walk (one lane per 256 edit steps of a traceback) -- the HIP-event bracket holds its chase, stitch and "
                 "the one-lane walk of refused alignments too; the walkers are > 95 % of it",
        "alg_bytes": st["seq_bases"] / 4.0 + st["d_steps"] / 8.0 + st["columns"] / 4.0,
        "formula": "seq_bases / 4 (operands) + d_steps / 8 (the path's move bit of every edit step) + columns / 4 (2-bit column kinds out); "
                   "the walkers' warm-up rows (32 per 256) are not algorithmic",
        "ms": st["traceback_ms"], "launches": st["traceback_launches"]}
    m["lq_score_kernel"] = {
        "label": "K12 lq_links + lq_score (low-quality-region rounds: links per run of regions, then scores / best links / walk per pile; "
                 "one HIP-event bracket over both, lq_score is ~85 % of it)",
        "alg_bytes": st["lq_aln_columns"] / 4.0 + st["lq_bases"] / 4.0 + float(st["lq_out"]),
        "formula": "lq_aln_columns / 4 (2-bit column kinds in) + lq_bases / 4 (2-bit candidate bases in) + lq_out (characters out)",
        "ms": st["lq_ms"], "launches": st["lq_launches"]}
    m["count_links_kernel"] = {
        "label": "K9 count_links (MSA link counting, wave per 32 columns)",
        "alg_bytes": 4.0 * st["tags"] + 12.0 * st["links"] + 8.0 * st["cells_msa"],
        "formula": "4 B per alignment tag in + 12 B per distinct link out + 8 B per MSA cell out",
        "ms": st["links_ms"], "launches": st["score_launches"]}
    m["score_seg_kernel"] = {
        "label": "K10 score_seg (segment-parallel scoring DP)",
        "alg_bytes": 20.0 * st["cells_msa"] + 12.0 * st["links"] + 20.0 * st["path_items"],
        "formula": "20 B per MSA cell (8 in, 12 out) + 12 B per link in + 20 B of column metadata",
        "ms": st["score_ms"], "launches": st["score_launches"]}
    # (one HIP-event bracket, two forms of the traceback: the one the run used is modelled)
    m.pop("ond_traceback_kernel" if st.get("tb_tasks", 0) > 0 else "tb_walk_kernel")
    return m


def committed_kernel_stats():
    """The newest profiles/rNN_bench_kernel_stats.txt (tools/rocprof_summary.py of `rocprofv3 --kernel-trace --stats -- python
    bench.py`): [(kernel name, calls, avg_us)] in the file's order (most GPU time first), and the file's name."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_kernel_stats.txt")))
    if not files:
        return [], None
    rows = []
    for ln in open(files[-1]):
        mt = re.match(r"^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
        if mt:
            rows.append((mt.group(1).strip(), int(mt.group(2)), float(mt.group(4))))
    return rows, os.path.relpath(files[-1], ROOT)


def committed_sq_ranking(top=6):
    """Whose the device is by the SQ counters: the first rows of the newest profiles/rNN_sq_wave_cycles_per_kernel.txt (one --pmc pass over
    SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY ..., tools/rocprof_sq_summary.py: every kernel alone on the device)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_sq_wave_cycles_per_kernel.txt")))
    if not files:
        return None
    rows = []
    for ln in open(files[-1]):
        mt = re.match(r"^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s+(\d+)\s*$", ln)
        if mt:
            rows.append({"kernel": mt.group(1).strip(), "launches": int(mt.group(2)), "wave_Mcycles": float(mt.group(3)),
                         "parked_pct": float(mt.group(4)), "stalled_at_issue_pct": float(mt.group(5)), "issuing_pct": float(mt.group(6))})
    rows.sort(key=lambda r_: -r_["wave_Mcycles"])
    return {"source": os.path.relpath(files[-1], ROOT), "by_wave_cycles": rows[:top],
            "note": "wave cycles of a kernel's wavefronts summed over its launches (quad-cycle counters), and where they go: parked on a wait "
                    "counter, stalled at instruction issue, issuing; the kernel the HBM roofline entry names is the one that leads this list too"}


_KERNEL_SOURCES = {   # the files a kernel's code lives in (its counter measurement is tied to them)
    "ond_forward_kernel": ("ond_kernels.hip", "nd_device.h"), "ond_traceback_kernel": ("ond_kernels.hip", "nd_device.h"),
    "tb_walk_kernel": ("ond_kernels.hip", "nd_device.h"),
    "count_links_kernel": ("msa_kernels.hip", "nd_device.h"), "score_seg_kernel": ("msa_kernels.hip", "nd_device.h"),
}


def kernel_source_sha16(kernel=None, read=None):
    """What a committed counter measurement of `kernel` is tied to: the sources its code lives in (every consensus kernel source for a
    kernel that is not listed).  `read`: file name -> bytes (tools/make_pmc_json.py hashes the tree a pass was measured on)."""
    h = hashlib.sha256()
    for f in _KERNEL_SOURCES.get(kernel, ("ond_kernels.hip", "lq_kernels.hip", "msa_kernels.hip", "nd_device.h")):
        if read:
            h.update(read(f))
        else:
            with open(os.path.join(ROOT, "nextdenovo_amd", "csrc", f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def committed_traffic(kernel, launches_per_step, config):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json, written by tools/make_pmc_json.py
    from two separate rocprofv3 --pmc runs) -- only if it was measured on these kernel sources and this workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pm = json.load(f)
    except (OSError, ValueError):
        return None, "no profiles/pmc_traffic.json"
    k = pm.get("kernels", {}).get(kernel)
    if k and k.get("source_sha16") != kernel_source_sha16(kernel):
        return None, "profiles/pmc_traffic.json: %s was measured on other sources (%s): not reported" % (kernel, k.get("source_sha16"))
    if not k or pm.get("config") != config or abs(k["launches_per_step"] - launches_per_step) > 0.5:
        return None, "profiles/pmc_traffic.json holds no matching entry for this kernel / workload"
    # MI355X_MICROARCH.md (HBM / rocprofv3): KB units; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes for wide reads -> x2
    return {"bytes": (2.0 * k["fetch_kb_per_launch"] + k["write_kb_per_launch"]) * 1024.0,
            "fetch_kb_raw": k["fetch_kb_per_launch"], "write_kb_raw": k["write_kb_per_launch"],
            "correction": "FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 bytes; MI355X_MICROARCH.md, HBM section), KB units",
            "source": pm.get("source")}, None


# ---- CPU baseline leg (the ONLY place bench.py touches oracle/) ---------------------------------
_REF = None
_CPU_CTX = None   # (read set, piles, max_lq_length, read type): inherited by the fork()ed workers


def _ref_worker(i):
    """One fork()ed worker = one `nextcorrect.py -p` worker (lib/nextcorrect.py:183-199):
    calls the compiled reference's nextCorrect() on one pile; returns the record's fingerprint and the CPU time the call took
    in this worker (process_time: the worker's own user + system time, not the wall of a pool that waits for its slowest pile)."""
    global _REF
    import ctypes as C
    from nextdenovo_amd import synth
    rs, piles, max_lq, rt = _CPU_CTX
    key = i
    seqs, st, en, mal = synth.pile_sequences(rs, piles[i])   # (the worker fetches its own sequences: lib/nextcorrect.py:189-193)
    mlq = min(en[0] // 2, max_lq)                            # lib/nextcorrect.py:188
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
    c0 = time.process_time()
    r = lib.nextCorrect(cs, (C.c_uint * n)(*st), (C.c_uint * n)(*en), n, mal, 500, 130, 4, mlq, 0.8, 0, 0, rt)
    cpu_s = time.process_time() - c0
    ln, ide = r.contents.len, r.contents.identity
    # what the parity block compares: length, the float32 identity bit for bit, md5 of the corrected bases
    # (error seeds -- len 2 / 3 / 4 -- carry no sequence: lib/nextcorrect.c:261-266, 2126)
    digest = hashlib.md5(C.string_at(r.contents.seq, ln)).hexdigest() if ln > 4 else ""
    bits = struct.unpack("<I", struct.pack("<f", ide))[0] if ln > 4 else 0
    lib.free_consensus_trimed(r)
    return key, ln, bits, digest, ide, cpu_s


def cpu_baseline(rs, piles, read_type, n_sample, max_lq, n_longest=16, n_runs=3):
    """Reference CPU path on a bounded sample of the same workload: the compiled reference's nextCorrect() over a fork pool of
    `cores` workers that STREAMS the piles (imap_unordered, one pile per task, longest seeds first), as lib/nextcorrect.py:232-235
    streams all of them -- with >= 16 piles per worker the pool's tail (its longest pile) amortises.  Every pile's CPU time is
    measured inside its worker, so the record carries both the makespan rate (`value` = bases / wall on `cores` workers) and the
    per-core rate (`per_core_measured` = bases / CPU-seconds; x cores = what a perfectly packed pool would do).
    The sample = every k-th pile + the `n_longest` longest seeds (where the device's wide tables, its int32 guard and its segment
    stitching matter); both parts count for the rates and for the parity block.
    Returns (record, {pile index: (len, identity bits, md5)})."""
    from multiprocessing import get_context
    from nextdenovo_amd import synth
    ref_so = os.path.join(ROOT, "oracle", "_ref", "nextcorrect.so")
    if not os.path.exists(ref_so) or not piles:
        return None, {}
    from nextdenovo_amd import hostinfo
    cores = max(1, min(hostinfo.effective_cpus(), 64))   # (the CPUs this process can have: a cgroup quota counts, see hostinfo.py)
    if n_sample <= 0:
        n_sample = min(len(piles), 16 * cores)
    step = max(1, len(piles) // n_sample)
    idx = list(range(0, len(piles), step))[:n_sample]
    by_len = sorted(range(len(piles)), key=lambda i: -int(piles[i]["recs"][0][3]))
    have = set(idx)
    idx += [i for i in by_len[:n_longest] if i not in have]
    idx.sort(key=lambda i: -int(piles[i]["recs"][0][3]))  # longest first: the pool's last tasks are its shortest

    global _CPU_CTX
    _CPU_CTX = (rs, piles, max_lq, read_type)
    ctx = get_context("fork")
    got = {}
    runs = []   # three runs, the median -- (wall, CPU-seconds) of each
    with ctx.Pool(cores) as pool:
        pool.map(_ref_worker, idx[-cores:], chunksize=1)  # warm: dlopen + page in, on the shortest piles
        for _run in range(max(1, n_runs)):
            t0 = time.perf_counter()
            for key, ln, bits, digest, ide, cpu_s in pool.imap_unordered(_ref_worker, idx, chunksize=1):
                got[key] = (ln, bits, digest, ide, cpu_s)
            runs.append((time.perf_counter() - t0, float(sum(g[4] for g in got.values()))))
    dt, cpu_s = sorted(runs)[len(runs) // 2]
    bases = int(sum(ln for ln, _b, _d, ide, _c in got.values() if ln > 4 and ide >= 0.8))
    slowest = max(g[4] for g in got.values())
    ref = {i: got[i][:3] for i in idx}
    _CPU_CTX = None
    per_core = bases / cpu_s if cpu_s > 0 else 0.0
    return {"value": bases / dt, "unit": "corrected bases/s", "cores": cores, "kind": "reference",
            "cores_note": "min(64, the CPUs this process can have: hardware threads %s, cgroup quota %s)" % (os.cpu_count(), hostinfo.cgroup_cpu_quota()),
            "wall_s": dt, "cpu_seconds": cpu_s, "per_core_measured": per_core, "per_core_measured_x_cores": per_core * cores,
            "runs": [{"wall_s": round(w, 3), "cpu_seconds": round(c, 2), "value": bases / w} for w, c in runs],
            "runs_note": "%d runs of the same sample on the same pool, `value` / `wall_s` / `cpu_seconds` = the median run (SURVEY.md section 8d)" % len(runs),
            "slowest_pile_cpu_s": slowest, "piles_per_worker": len(idx) / cores,
            "sample": "%d of %d piles (every %d-th + the %d longest seeds), %d corrected bases; compiled reference nextcorrect.so, fork pool "
                      "of %d workers streaming one pile per task, longest first (lib/nextcorrect.py:232-235): %.2f s wall, %.1f CPU-seconds "
                      "summed over the piles (process_time inside the workers), slowest pile %.2f CPU-s; `value` = bases / wall, "
                      "`per_core_measured` = bases / CPU-seconds" % (len(idx), len(piles), step, n_longest, bases, cores, dt, cpu_s, slowest)}, ref


def parity_block(ref, gpu_full, piles):
    """Compare the device's records of the CPU-sample piles (taken from a whole-batch call, the very call shape
    the timed steps make) with the compiled reference's: length, float32 identity bits, md5 of the bases.  Seeds the
    reference itself could not hold in memory (its len 3, lib/nextcorrect.c:2254-2261) are counted, not compared."""
    bad, ref_oom, from_file = [], 0, 0
    for i, (ln, bits, digest) in sorted(ref.items()):
        g_ln, g_ide, g_seq = gpu_full[i]
        if ln == 3:
            ref_oom += 1
            continue
        if ln > 4 or g_ln > 4:
            g_bits = struct.unpack("<I", struct.pack("<f", g_ide))[0] if g_ln > 4 else 0
            if g_seq is None:   # (a record the stage's output filter dropped -- lib/nextcorrect.py:236 -- is not in cns.fasta)
                same = (ln, bits) == (g_ln, g_bits)
            else:
                g_dig = hashlib.md5(g_seq).hexdigest() if g_ln > 4 else ""
                same = (ln, bits, digest) == (g_ln, g_bits, g_dig) and len(g_seq) == g_ln
                from_file += 1
        else:
            same = ln == g_ln  # error seeds: the convention code only (2 uncorrectable, 3 memory, 4 all clipped)
        if not same:
            bad.append({"pile": i, "seed": int(piles[i]["seed"]), "ref_len": ln, "gpu_len": g_ln})
    lens = [int(piles[i]["recs"][0][3]) + 1 for i in ref]
    depth = [int(piles[i]["recs"].shape[0]) for i in ref]
    return {"piles": len(ref), "mismatch": len(bad), "against": "oracle/_ref/nextcorrect.so (compiled reference)",
            "compared": "len, float32 identity bits, md5(seq); the sequences are the bytes the last timed step wrote into its cns.fasta "
                        "(%d records read back from the file)" % from_file if from_file else "len, float32 identity bits, md5(seq)",
            "longest_seed": max(lens) if lens else 0,
            "seeds_ge_100kb": sum(1 for x in lens if x >= 100000), "deepest_pile": max(depth) if depth else 0,
            "error_seeds": sum(1 for v in ref.values() if v[0] <= 4), "reference_out_of_memory_seeds": ref_oom,
            "device_out_of_memory_seeds": sum(1 for i in ref if gpu_full[i][0] == 3), "mismatches": bad[:8]}


def cpu_baseline_overlap(rs_dev, preset, device_records=None, sort_k=0):
    """Reference overlapper (oracle/_ref/minimap2-nd --step 1 -I 3G -t cores, the seed x seed job of nextDenovo:456-464) on the
    same read set: the whole all-vs-all job, wall time including its index build; with `device_records` (what the device's stage
    produced for the same job) also whether the two `.ovl` byte streams are the same."""
    import hashlib
    import subprocess
    import tempfile
    import numpy as np
    from nextdenovo_amd import ovl, overlap
    exe = os.path.join(ROOT, "oracle", "_ref", "minimap2-nd")
    if not os.path.exists(exe):
        return None
    from nextdenovo_amd import hostinfo
    cores = max(1, min(hostinfo.effective_cpus(), 64))
    wd = tempfile.mkdtemp(prefix="ndbench")
    p = os.path.join(wd, "reads.2bit")
    ovl.write_2bit(p, rs_dev.ids, rs_dev.lens, rs_dev.words, rs_dev.word_off)
    out = os.path.join(wd, "ref.ovl")
    t0 = time.perf_counter()
    subprocess.run([exe, "--step", "1", "-I", "3G", "-t", str(cores), "-x", preset, p, p, "-o", out], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    nbytes = os.path.getsize(out)
    bases = int(rs_dev.lens.sum())
    res = {"value": bases / dt, "unit": "query bases/s", "cores": cores, "kind": "reference", "wall_s": dt, "ovl_bytes": nbytes,
           "sample": "whole read set (%d reads, %d bases) all-vs-all, compiled reference minimap2-nd --step 1 -I 3G -t %d -x %s"
                     % (len(rs_dev), bases, cores, preset)}
    if device_records is not None:
        h = hashlib.md5()
        with open(out, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        mine = overlap.encode(device_records, np.zeros(2, dtype=np.uint32))
        res["device_ovl_bytes"] = len(mine)
        res["device_ovl_identical"] = len(mine) == nbytes and hashlib.md5(mine).hexdigest() == h.hexdigest()
    # sort_align on the CPU (the wall of raw_align + sort_align + seed_cns): the compiled reference's ovl_sort on
    # the .ovl the reference overlapper just wrote, as nextDenovo:344-354 runs it (-k as lib/config_parser.py:44)
    srt_exe = os.path.join(ROOT, "oracle", "_ref", "ovl_sort")
    if sort_k and os.path.exists(srt_exe):
        try:
            ids = np.asarray(rs_dev.ids, dtype=np.int64)
            ln = np.asarray(rs_dev.lens, dtype=np.int64)
            cnt = (ln + 15) >> 4
            start = np.zeros(ids.size + 1, dtype=np.int64)
            np.cumsum(cnt + 2, out=start[1:])
            with open(os.path.join(wd, ".reads.idx"), "w") as f:   # id, byte offset of the read's first word, length (util/seq_dump.c:39)
                f.write("".join("%d\t%d\t%d\n" % (i, 2 + 4 * (st + 2), l_) for i, st, l_ in zip(ids.tolist(), start[:-1].tolist(), ln.tolist())))
            with open(os.path.join(wd, "in.fofn"), "w") as f:
                f.write(out + "\n")
            t0 = time.perf_counter()
            subprocess.run([srt_exe, "-m", "40g", "-t", str(max(2, cores)), "-k", str(sort_k), "-i", os.path.join(wd, ".reads.idx"), "-o", "ref.sorted.ovl",
                            "in.fofn"], check=True, cwd=wd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            res["ovl_sort"] = {"wall_s": time.perf_counter() - t0, "threads": max(2, cores), "kind": "reference",
                               "sorted_ovl_bytes": os.path.getsize(os.path.join(wd, "ref.sorted.ovl")),
                               "command": "oracle/_ref/ovl_sort -m 40g -t %d -k %d -i .reads.idx -o ref.sorted.ovl in.fofn" % (max(2, cores), sort_k)}
            if device_records is not None:   # the device's sort of the device's records against the reference chain's bytes
                seed_len = np.zeros(int(ids.max()) + 1 if ids.size else 0, dtype=np.uint32)
                seed_len[ids] = ln
                srt, bl, _ = overlap.sort_overlaps([device_records], seed_len, int(ln.min()) if ln.size else 0, sort_k, 300)
                mine = overlap.encode(srt, np.zeros(2, dtype=np.uint32))
                with open(os.path.join(wd, "ref.sorted.ovl"), "rb") as f:
                    theirs = f.read()
                res["ovl_sort"]["device_sorted_ovl_identical"] = mine == theirs
                res["ovl_sort"]["blacklisted_reads"] = len(bl)
        except (OSError, subprocess.CalledProcessError) as e_:
            res["ovl_sort"] = {"error": repr(e_)}
    import shutil
    shutil.rmtree(wd, ignore_errors=True)
    return res


def host_description():
    """What the step's host phases ran on: the figure moves with the box, so the line says which box."""
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = None
    from nextdenovo_amd import hostinfo
    return {"cpu_model": model, "cpu_count": os.cpu_count(), "cpus_allowed": aff, "cgroup_cpu_quota": hostinfo.cgroup_cpu_quota(),
            "effective_cpus": hostinfo.effective_cpus()}


def host_snapshot():
    """Load average and the GPU's clocks / busy figure as sysfs shows them (no tool is run): taken before and after the timed steps."""
    import glob
    snap = {}
    try:
        snap["loadavg"] = [round(x, 1) for x in os.getloadavg()]
    except OSError:
        pass
    from nextdenovo_amd import hostinfo
    th = hostinfo.throttle_stat()
    if th:
        snap["cgroup_periods_throttled_usec"] = list(th)
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))[:1]:
        for key, fn in (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk"), ("gpu_busy_percent", "gpu_busy_percent")):
            try:
                with open(os.path.join(card, fn)) as f:
                    txt = f.read().strip()
            except OSError:
                continue
            if key == "gpu_busy_percent":
                snap[key] = txt
            else:  # the active level carries a '*'
                act = [ln.split(":", 1)[1].strip().rstrip("*").strip() for ln in txt.splitlines() if ln.strip().endswith("*")]
                snap[key] = act[0] if act else txt.splitlines()[-1].strip() if txt else ""
    return snap


def reduce_over_ranks(dist, torch, bases: int, dt: float, device, seeds: int = 0):
    """The path's only collective: sum of corrected bases (and corrected seeds), max of wall
    time.  RCCL over xGMI on the GPU box ("nccl" backend), gloo in the CPU tests."""
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    b = torch.tensor([bases, seeds], dtype=torch.int64, device=device)
    dist.all_reduce(b, op=dist.ReduceOp.SUM)
    reduce_over_ranks.seeds = int(b[1].item())
    return int(b[0].item()), float(t.item())


def gather_rank_stats(dist, torch, device, values):
    """Every rank's (wall s, overlap s, sort + piles s, consensus s, piles, alignment columns): one all_gather of six doubles, so
    that imbalance between the seed files and the index rebuilds of the mirror jobs show in the one line rank 0 prints."""
    mine = torch.tensor(values, dtype=torch.float64, device=device)
    parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    keys = ("wall_s", "overlap_s", "sort_piles_s", "consensus_s", "piles", "columns")
    return [{k: (float(v) if k.endswith("_s") else int(v)) for k, v in zip(keys, p.tolist())} for p in parts]


def shard_of_rank(world: int, rank: int, seed_files: int, shard: int):
    """(seed_cutfiles, the seed file this rank corrects): rank r of N takes seed file r of N; one GPU with --seed-files M
    takes seed file --shard of M."""
    if world > 1:
        return world, rank
    m = max(1, seed_files)
    return m, shard % m


def make_workload(args):
    """(name, ReadSet, words, word_off, lens, preset, read_type, max_lq_length, sort -k, genome size, depth, profile):
    the same read set on every rank."""
    from nextdenovo_amd import synth
    cfg = dict(synth.CONFIGS[args.config])
    custom = bool(args.genome_size or args.depth or args.profile)
    depth = args.depth or cfg["depth"]
    profile = args.profile or cfg["profile"]
    if custom:
        gs = int(args.genome_size or 4.6e6)
        genome = synth.make_genome(gs, seed=42)
        name = "synthetic uniform %.2f Mb, %gx %s reads (lognormal mu %.2f sigma %.2f)" % (gs / 1e6, depth, profile, cfg["mu"], cfg["sigma"])
    else:
        genome = cfg["genome"]()
        name = cfg["name"]
    if args.config == 2:  # the round-1 read set, generated in one stream
        rs = synth.simulate_reads(genome, depth, profile, seed=43, mu=cfg["mu"], sigma=cfg["sigma"], max_len=cfg["max_len"])
        words, word_off, lens = synth.pack_db(rs)
    else:
        rs, words, word_off, lens = synth.simulate_reads_mp(genome, depth, profile, 43 + args.config, cfg["mu"], cfg["sigma"], cfg["max_len"])
    d = int(round(depth))
    sort_k = (d - 2) if d <= 30 else min(d - 5, 40)  # lib/config_parser.py:44
    preset = "ava-hifi" if profile == "hifi" else "ava-ont" if profile == "ont" else "ava-pb"
    max_lq = cfg["max_lq"] if not args.profile else (10000 if profile == "ont" else 1000)  # lib/config_parser.py:217-221
    return name, rs, words, word_off, lens, preset, {"ont": 1, "clr": 2, "hifi": 3}[profile], max_lq, sort_k, genome.size, depth, profile


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("NDGPU_DEVICE", str(local_rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    from nextdenovo_amd import api, stage, synth

    dist = None
    torch = None
    # NDGPU_BENCH_FORCE_DIST=1: the process group, the barrier and the RCCL all-reduce / all-gather of an N > 1 run with ONE rank --
    # the rehearsal of the collective leg that a one-GPU box allows (`torch.distributed.run --nproc-per-node 1`)
    if world > 1 or os.environ.get("NDGPU_BENCH_FORCE_DIST"):
        import torch
        import torch.distributed as dist
        # ("nccl" IS RCCL on ROCm.  NDGPU_BENCH_DIST_BACKEND=gloo is the test hook of tests/test_bench_cpu.py: main() itself, N = 2, on
        # a machine without a GPU)
        dist_backend = os.environ.get("NDGPU_BENCH_DIST_BACKEND", "nccl")
        if dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(dist_backend)

    dist_dev = "cuda" if (dist is None or os.environ.get("NDGPU_BENCH_DIST_BACKEND", "nccl") == "nccl") else "cpu"
    analytic = args.analytic_piles or args.no_overlap
    if args.host_threads <= 0:  # the ranks of one node share its cores
        from nextdenovo_amd import hostinfo
        args.host_threads = max(4, hostinfo.effective_cpus() // max(1, world))
    t_gen = time.perf_counter()
    name, rs, words, word_off, lens, preset, read_type, max_lq, sort_k, genome_size, depth, profile = make_workload(args)
    n_files, my_file = shard_of_rank(world, rank, args.seed_files or world, args.shard)
    t_gen = time.perf_counter() - t_gen

    db = api.ReadDB(words, word_off, lens)  # every read, both strands, resident in HBM from here on
    exchange = None
    if world > 1 and not args.no_exchange:
        # the ranks of this node hand seed x seed job records to one another through a node-local directory, as the reference hands
        # the .ovl file of a pair to its second reader with `ln -sf` (nextDenovo:455-459): a pair is mapped once per node
        xdir = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", "ndgpu_bench_%s" % os.environ.get("MASTER_PORT", "0"))
        if rank == 0:   # a run that died may have left files of ITS job layout here: start from an empty directory
            import shutil
            shutil.rmtree(xdir, ignore_errors=True)
            os.makedirs(xdir, exist_ok=True)
        dist.barrier()      # (nobody writes before the directory is clean)
        exchange = stage.Exchange(xdir, rank)
    sh = stage.Shard(words, word_off, lens, preset=preset, seed_cutoff=1000, read_cutoff=500, n_seed_files=n_files, sort_k=sort_k,
                     exchange=exchange)
    # --producers 2: a second Shard for a second producer of piles (stage.StagePipeline).  Under the consensus kernels a stage's many
    # short kernels wait in line -- one stage takes ~0.45 s where it takes 0.09 s alone -- and the line waits ~160 ms per step for piles;
    # with two stages side by side it waits 70 ms and the step is the same 560-570 ms (profiles/r06_pipeline_ab.txt): the device is
    # the bound, so one producer is the default.  One GPU only (the ranks of a node step their hand-over together).
    # N > 1: a rank's stage takes two to three times as long beside its consensus as alone (80 -> 280 ms for a rank of 2, 40 -> 110 for a
    # rank of 8) and is what the rank's line waits for; two producers -- each Shard with an Exchange object of its own on the node's
    # directory, the steps numbered here -- keep it fed.  Only where the rank has CPUs for them.
    n_prod = max(1, args.producers)
    if args.producers <= 0 and world > 1:
        from nextdenovo_amd import hostinfo as _hi
        n_prod = 2 if _hi.effective_cpus() // world >= 6 else 1
    if args.no_pipeline or args.no_overlap:
        n_prod = 1
    shards = [sh]
    if n_prod > 1 and exchange is not None:
        exchange.lookahead, exchange.KEEP_STEPS = 2, 7   # (StagePipeline.ahead; see Exchange.__init__)
    for _ in range(n_prod - 1):
        ex_b = stage.Exchange(exchange.dir, rank, lookahead=2) if exchange is not None else None
        shards.append(stage.Shard(words, word_off, lens, preset=preset, seed_cutoff=1000, read_cutoff=500, n_seed_files=n_files, sort_k=sort_k,
                                  exchange=ex_b))
    last_lock = __import__("threading").Lock()
    a_piles = a_recs = a_off = a_names = None
    if analytic:  # piles from the true read positions, for the seeds of this rank's seed file
        a_piles = synth.build_piles(rs, seed_cutoff=1000, seed_ids=[int(i) for i in sh.seed_ids[my_file]])
        a_recs, a_off = synth.flatten_piles(a_piles)
        a_names = [int(p_["seed"]) for p_ in a_piles]
    last = {}
    cns_wall = [0.0]

    def sync():
        if dist is not None:
            dist.barrier()
            if dist_dev == "cuda":
                torch.cuda.synchronize()

    # The stage's output is part of the stage (lib/nextcorrect.py:236-260): every step writes its cns.fasta + cns.fasta.idx, record
    # for record as the reference's loop prints them, into a scratch directory (removed at the end).
    import tempfile
    out_dir = tempfile.mkdtemp(prefix="ndgpu_bench_cns_")
    fa_path = os.path.join(out_dir, "cns.%d.fasta" % rank)
    write_wall = [0.0]
    stream_wall = [0.0]
    fasta_bytes = [0]
    last_res = []

    # The stage of the NEXT seed file begins while this one's consensus runs (a node has more seed files than GPUs: nextDenovo:344-354
    # runs sort_align / seed_cns once per seed file, and the raw_align jobs of the next one are independent of this one's consensus):
    # its overlap, sort and pile-admission kernels fill what the consensus kernels leave of the device, its host work what the
    # contexts' host phases leave of the CPUs; and two consensus calls are in flight, the tail of one under the main phases of the
    # next (stage.StagePipeline -- the product's `correct_stage` command runs its seed files through the same class).  Every timed
    # step still runs BOTH halves inside the timed region: the first timed step computes its own piles (nothing is prefetched across
    # the start of the clock), the last one prefetches nothing.
    import threading
    pipeline = {"on": not (args.no_pipeline or args.no_overlap), "depth": 1, "wait_s": 0.0}

    def get_piles(_k=None):
        """(records, pile offsets, names) of a step."""
        if args.no_overlap:
            return a_recs, a_off, a_names
        sub, off, seeds, n_bl = shards[(_k or 0) % len(shards)].piles(my_file, step=None if _k is None else step_base[0] + 1 + _k)
        with last_lock:
            last.update(sub=sub, off=off, seeds=seeds, n_bl=n_bl)
        return (a_recs, a_off, a_names) if analytic else (sub, off, seeds)

    acc_lock = threading.Lock()

    def consensus(r_, o_, names, path):
        """The consensus half of a step: every admitted pile through the library, cns.fasta + .idx written as the sub-batches finish."""
        t_c = time.perf_counter()
        n_ok = b_ok = 0
        if args.lengths_only:
            res = db.correct_piles(r_, o_, read_type=read_type, max_lq_length=max_lq, host_threads=args.host_threads, lengths_only=True)
            t_w = time.perf_counter()
            with acc_lock:
                cns_wall[0] += t_w - t_c
        else:
            with open(path, "wb") as OUT, open(path + ".idx", "wb") as IDX:
                t_lib = [0.0, 0.0]
                res = db.correct_piles(r_, o_, read_type=read_type, max_lq_length=max_lq, host_threads=args.host_threads,
                                       fasta=(OUT, IDX, names, 500, 0.8), lib_wall=t_lib)
                t_w = t_c + t_lib[0]
                with acc_lock:
                    cns_wall[0] += t_lib[0]
                    stream_wall[0] += t_lib[1]
                    fasta_bytes[0] = OUT.tell()
        # accepted records exactly as lib/nextcorrect.py:236 (len >= min_len_seed(=seed_cutoff/2), identity >= ratio)
        for ln, ide in res:
            if ln >= 500 and ln > 4 and ide >= 0.8:
                n_ok += 1
                b_ok += ln
        with acc_lock:
            write_wall[0] += time.perf_counter() - t_w
        return b_ok, n_ok, res, path, time.perf_counter()

    def step():
        r_, o_, names = get_piles()
        b_ok, n_ok, res, _, _ = consensus(r_, o_, names, fa_path)
        last_res[:] = [res]
        return b_ok, n_ok

    step_base = [0]   # the hand-over's step number of the last warm-up step: timed step k is step_base + 1 + k on every Shard of the rank
    for _ in range(args.warmup):
        step()
    step_base[0] = args.warmup
    cns_wall[0] = 0.0
    write_wall[0] = 0.0
    stream_wall[0] = 0.0
    api.reset_stats()
    if not args.no_overlap:
        from nextdenovo_amd import overlap as _ovl
        _ovl.pool_calls(reset=True)
    for sh_ in shards:
        for k in sh_.stats:
            sh_.stats[k] = 0
    host0 = host_snapshot()
    sync()
    t0 = time.perf_counter()
    bases = n_ok = 0
    step_s = []
    depth = min(max(1, args.consensus_depth), 4) if (pipeline["on"] and args.steps > 1) else 1
    if not pipeline["on"]:
        for k_step in range(args.steps):
            t_s = time.perf_counter()
            b, n = step()
            step_s.append(time.perf_counter() - t_s)
            bases += b
            n_ok += n
    else:
        line = stage.StagePipeline(get_piles, lambda k, piles: consensus(piles[0], piles[1], piles[2], fa_path + (".%d" % (k % 4))),
                                   depth=depth, prefetch=True, producers=len(shards))
        ends = []
        for _k, (b, n, res, path, _t), t_end in line.run(range(args.steps)):
            ends.append(t_end)
            bases += b
            n_ok += n
            if t_end >= max(ends):   # (the parity block reads the file of the call that ended last)
                last_res[:] = [res]
                last["fa_path"] = path
        # a step's time = from the end of the call before it to its own end, in the order the calls ended (two are in flight)
        t_last = t0
        for t_end in sorted(ends):
            step_s.append(t_end - t_last)
            t_last = t_end
        pipeline.update(on=line.prefetch, depth=line.depth, wait_s=line.wait_s)
    for sh_ in shards[1:]:   # (the stage times of every producer's Shard, summed)
        for k in sh_.stats:
            sh.stats[k] += sh_.stats[k]
        if sh.ovl_stats is None:
            sh.ovl_stats = sh_.ovl_stats
    sync()
    dt = time.perf_counter() - t0
    host1 = host_snapshot()
    st = api.stats()
    pool_timed = None
    if not args.no_overlap:
        pool_timed = _ovl.pool_calls()

    total_bases, max_dt, total_seeds = bases, dt, n_ok
    per_rank = None
    if dist is not None:
        total_bases, max_dt = reduce_over_ranks(dist, torch, bases, dt, dist_dev, n_ok)
        total_seeds = reduce_over_ranks.seeds
        per_rank = gather_rank_stats(dist, torch, dist_dev, [dt, sh.stats.get("overlap_s", 0.0) if not args.no_overlap else 0.0,
                                                           (sh.stats.get("sort_s", 0.0) + sh.stats.get("assemble_s", 0.0)) if not args.no_overlap else 0.0,
                                                           cns_wall[0], float(st["piles"]), float(st["path_items"])])

    if rank == 0:
        if analytic:
            piles, recs, off = a_piles, a_recs, a_off
        else:
            recs, off = last["sub"], last["off"]
            piles = [{"seed": int(last["seeds"][i]), "recs": recs[int(off[i]):int(off[i + 1])]} for i in range(last["seeds"].size)]
        # Roofline: the kernel that leads the GPU time of the committed rocprofv3 kernel trace of this command (profiles/), priced
        # with the algorithmic bytes of SURVEY.md section 8d from this run's counters over its HIP-event launch time; the other
        # modelled kernels beside it.
        models = kernel_models(st)
        stat_rows, stat_file = committed_kernel_stats()
        dom = None
        # (a modelled kernel may appear as several instantiations -- K7 with and without checkpoints -- that share one HIP-event bracket
        # and one model: their rows are taken together, calls-weighted)
        agg = {}
        for kname, calls, avg_us in stat_rows:
            for key in models:
                if kname.startswith(key):
                    a_ = agg.setdefault(key, [0, 0.0])
                    a_[0] += calls
                    a_[1] += calls * avg_us
        for kname, calls, avg_us in stat_rows:   # (rows in order of GPU time)
            for key in models:
                if kname.startswith(key):
                    dom = (key, kname, agg[key][0], agg[key][1] / max(1, agg[key][0]))
                    break
            if dom:
                break
        if dom is None:  # no committed profile: the modelled kernel with the most event time in this run
            key = max(models, key=lambda k_: models[k_]["ms"])
            dom = (key, key, 0, 0.0)

        def entry(key, rocprof_avg_us=None):
            md = models[key]
            launches = max(1, int(md["launches"]))
            per_launch = md["alg_bytes"] / launches
            avg_ms = md["ms"] / launches
            ach = per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            e = {"kernel": key, "what": md["label"], "achieved": ach, "frac": ach / 8000.0, "alg_bytes_per_launch": per_launch,
                 "alg_bytes_formula": md["formula"], "avg_launch_ms": avg_ms, "launches": launches,
                 "launches_per_step": launches / args.steps}
            if rocprof_avg_us:
                e["rocprof_avg_launch_ms"] = rocprof_avg_us / 1e3
                e["frac_at_rocprof_avg"] = per_launch / (rocprof_avg_us * 1e-6) / 1e9 / 8000.0
            return e
        same_workload = args.config == 2 and not args.genome_size and world == 1 and n_files == 1 and not analytic  # what the committed trace ran
        dom_entry = entry(dom[0], (dom[3] or None) if same_workload else None)
        traffic, traffic_note = None, "analytic / sharded runs carry no committed counter figure"
        if not analytic and world == 1 and n_files == 1:
            traffic, traffic_note = committed_traffic(dom[0], dom_entry["launches_per_step"], args.config)
        others = {}
        rp = {key: v[1] / max(1, v[0]) for key, v in agg.items()}
        for key in models:
            if key != dom[0]:
                others[key] = entry(key, rp.get(key) if same_workload else None)
        peak = 8000.0
        n_cols = max(1, st["path_items"])
        out = {
            "metric": "corrected bases/sec",
            "value": total_bases / max_dt,
            "unit": "bases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": max_dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": "BASELINE config %d: %s, seed_cutoff 1k, -x %s, ovl_sort -k %d" % (args.config, name, preset, sort_k),
                       "reads": len(rs), "read_bases": int(lens.sum()), "genome_size": int(genome_size), "depth": depth, "profile": profile,
                       "seed_files": n_files, "sharding": "seed file r of %d on rank r (one read set; util/seq_dump.c:87-92 dealing)" % n_files
                       if world > 1 else "seed file %d of %d on the one GPU" % (my_file, n_files),
                       "piles_rank0": len(piles), "overlaps_rank0": int(recs.shape[0]), "corrected_seeds_per_step": total_seeds / args.steps,
                       "piles_from": "analytic (true read positions)" if analytic else
                       "the step's own overlap -> ovl_sort -> pile assembly chain on the device",
                       "longest_chain_bound": "none since round 2: the scoring DP and the best_pp walk are cut into 1024-column segments "
                                              "scored at the same time (DESIGN section 5); a seed's length no longer bounds a step",
                       "datagen_s": round(t_gen, 1)},
            "roofline": dict(dom_entry, **{
                "bound": "hbm", "peak": 8000.0, "unit": "GB/s",
                "dominant_by": ("%s: `%s` leads the GPU time of the committed kernel trace" % (stat_file, dom[1])) if stat_file else
                               "no committed kernel trace: the modelled kernel with the most HIP-event time in this run",
                "traffic": traffic["bytes"] if traffic else None,
                "traffic_unit": "HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes)",
                "traffic_detail": traffic if traffic else traffic_note,
                "traffic_over_algorithmic": (traffic["bytes"] / dom_entry["alg_bytes_per_launch"]) if traffic and dom_entry["alg_bytes_per_launch"] else None,
                "cells_per_s_k7": st["cells"] / (st["forward_ms"] * 1e-3) if st["forward_ms"] > 0 else 0.0,
                "note": "every consensus kernel is a chain of dependent steps per alignment / pile / column, bound by instruction issue and "
                        "latency, not by bytes (SURVEY.md section 8d expects << 1 % of the HBM roofline); launch times are HIP-event "
                        "brackets on the kernel's own stream with up to 8 contexts' launches in flight at once, `rocprof_avg_launch_ms` "
                        "is the same kernel's average in the committed trace",
                "timed_step_note": ("--lengths-only: the library computes every corrected sequence, the hand-over of the bytes and the "
                                    "cns.fasta write are skipped" if args.lengths_only else
                                    "every step takes the corrected sequences from the library and writes cns.fasta + cns.fasta.idx "
                                    "(lib/nextcorrect.py:236-260) inside the timed region"),
                "device_load_by_sq_counters": committed_sq_ranking(),
                "other_kernels": others}),
            "counters": {k: st[k] for k in ("tasks", "wide_tasks", "cells", "d_steps", "trace_words", "lq_rounds", "lq_declined", "max_band", "piles", "tags",
                                            "cells_msa", "links", "path_items", "score_segments", "score_repairs", "score_slow_piles",
                                            "lq_jobs", "lq_repairs", "lq_columns", "tb_tasks", "tb_walkers", "tb_fallbacks")},
            "kernel_ms": {k: round(st[k], 2) for k in ("forward_ms", "traceback_ms", "tags_ms", "links_ms", "score_ms",
                                                       "backtrack_ms", "extract_ms", "lq_ms")},
            "consensus_ms_per_step": cns_wall[0] / args.steps * 1e3,
            "pipeline": {"next_stage_begins_during_consensus": bool(pipeline["on"]), "consensus_calls_in_flight": pipeline.get("depth", 1), "piles_producers": len(shards),
                         "wait_for_prefetched_piles_ms_per_step": pipeline["wait_s"] / args.steps * 1e3,
                         "note": "with pipelining the overlap / sort / pile-admission stage of step k + 1 runs on its own host thread and "
                                 "streams while the consensus of step k holds the device; the first timed step computes its piles itself and "
                                 "the last one prefetches nothing, so K overlap stages and K consensus stages lie inside the timed region; "
                                 "the per-stage times then add up to more than ms_per_step (--no-pipeline: one stage after the other)"},
            "fasta_write": {"ms_per_step": (write_wall[0] + stream_wall[0]) / args.steps * 1e3, "bytes_per_step": fasta_bytes[0],
                            "of_which_after_the_last_sub_batch_ms": write_wall[0] / args.steps * 1e3,
                            "how": "records written sub-batch by sub-batch from the library's completion callback while later sub-batches "
                                   "are on the device (ndgpu_correct_piles_stream; lib/nextcorrect.py:232-260 prints each seed as its worker "
                                   "returns it); the file close follows the call",
                            "included_in_value": not args.lengths_only},
            # every timed step by itself (rank 0), so that a slow box, a cold start and noise can be told apart; `value` stays
            # total bases / total wall of the K steps (the launch contract), the median is beside it
            "step_ms": {"list": [round(x * 1e3, 1) for x in step_s], "min": round(min(step_s) * 1e3, 1),
                        "median": round(sorted(step_s)[len(step_s) // 2] * 1e3, 1),
                        "p90": round(sorted(step_s)[min(len(step_s) - 1, int(0.9 * len(step_s)))] * 1e3, 1),
                        "max": round(max(step_s) * 1e3, 1),
                        "value_at_median_step": (bases / args.steps) / sorted(step_s)[len(step_s) // 2]},
            "host": dict(host_description(), host_threads=args.host_threads, contexts=int(os.environ.get("NDGPU_CONTEXTS", "0")) or "default",
                         before=host0, after=host1),
            # buffers (re)allocated inside the timed steps: while batches ran (these stall every context) / between batches
            "allocations": {"in_step": int(st["allocs"]), "in_step_ms": round(st["alloc_ms"], 1), "between_batches": int(st["level_allocs"]),
                            "between_batches_ms": round(st["level_ms"], 1)},
        }
        if per_rank is not None:
            out["per_rank"] = per_rank   # over the K timed steps; rank r = seed file r
        if world > 1:
            pl = stage.plan(n_files, len(sh.part_ids))
            out["config"]["raw_align_jobs"] = {
                "exchange": exchange is not None,
                "per_rank_jobs_computed": [r_["jobs_computed_with_exchange" if exchange is not None else "jobs_computed_alone"] for r_ in pl],
                "per_rank_index_builds": [r_["index_builds_with_exchange" if exchange is not None else "index_builds_alone"] for r_ in pl],
                "rank0_exchange": ({k_: sum(sh_.exchange.stats[k_] for sh_ in shards if sh_.exchange is not None) for k_ in exchange.stats}
                                   if exchange is not None else None),
                "rank0_piles_producers": len(shards),
                "note": "a seed x seed pair is mapped once per node and handed over through /dev/shm (nextDenovo:455-459 does it with "
                        "`ln -sf`); --no-exchange: every rank maps the mirrors it needs itself"}
        if not args.no_overlap:
            q_bases = int(lens.sum())
            o_ms = sh.stats["overlap_s"] / args.steps * 1e3
            out["overlap"] = {"included_in_value": True, "ms_per_step": o_ms, "jobs_per_step": sh.stats["jobs"] / args.steps,
                              "records_per_step": sh.stats["records"] / args.steps,
                              "value": q_bases / (o_ms * 1e-3) if o_ms > 0 else 0.0,
                              "unit": "query bases/s (this rank's raw_align jobs, index builds included)",
                              "sort": {"ms_per_step": sh.stats["sort_s"] / args.steps * 1e3, "blacklisted": int(last.get("n_bl", 0)), "k": sort_k},
                              "pile_assembly_ms_per_step": sh.stats["assemble_s"] / args.steps * 1e3,
                              # hipMalloc / hipFree calls of the overlap library's block pool inside the timed steps (each stalls every stream)
                              "pool_calls": {"n": pool_timed[0], "ms": round(pool_timed[1] * 1e3, 1)}}
            ost = sh.ovl_stats
            if ost:
                out["overlap"]["kernel_ms_last_index"] = {k: round(ost[k], 3) for k in ("sketch_ms", "index_sort_ms", "seed_ms", "sort_ms",
                                                                                         "exact_sort_ms", "chain_ms", "hits_ms") if k in ost}
                out["overlap"]["counters_last_index"] = {k: int(ost[k]) for k in ("bases_sketched", "minimizers", "anchors", "tie_reads",
                                                                                  "chain_cells", "chains", "overlaps") if k in ost}
        parity_fail = False
        if not args.no_cpu_baseline and world == 1:  # the CPU leg runs at N = 1 only
            out["cpu_baseline"], ref = cpu_baseline(rs, piles, read_type, args.cpu_sample, max_lq)
            fa_path = last.get("fa_path", fa_path)   # (the file of the step that ended last)
            if ref and not args.lengths_only and last_res and os.path.exists(fa_path):
                # what the LAST TIMED STEP wrote: (len, identity) of every pile from the library's records, the bases from the
                # step's cns.fasta (a record the output filter dropped has no bases there: length and identity only)
                by_name = {}
                with open(fa_path, "rb") as f:
                    blob = f.read()
                at = 0
                while at < len(blob):
                    e1 = blob.index(b"\n", at)
                    e2 = blob.index(b"\n", e1 + 1)
                    by_name[int(blob[at + 1:e1].split()[0])] = blob[e1 + 1:e2]
                    at = e2 + 1
                full = [(ln, ide, by_name.get(int(piles[i]["seed"]))) for i, (ln, ide) in enumerate(last_res[0])]
                out["parity"] = parity_block(ref, full, piles)
                out["parity"]["fasta_records"] = len(by_name)
                parity_fail = out["parity"]["mismatch"] != 0
            elif ref:  # one more (untimed) whole-batch call that keeps the sequences: what the timed steps computed
                full = db.correct_piles(recs, off, read_type=read_type, max_lq_length=max_lq, host_threads=args.host_threads)
                out["parity"] = parity_block(ref, full, piles)
                parity_fail = out["parity"]["mismatch"] != 0
            if not args.no_overlap and n_files == 1 and args.config in (2, 3) and not args.no_overlap_baseline:
                from nextdenovo_amd import overlap
                files = sh.overlaps(my_file)   # (untimed) the records of the one raw_align job of this layout: seed file x itself
                out["overlap"]["cpu_baseline"] = cpu_baseline_overlap(
                    overlap.ReadSet(np.arange(len(rs), dtype=np.uint32), lens, words, word_off), preset, files[0] if len(files) == 1 else None, sort_k=sort_k)
                osrt = (out["overlap"]["cpu_baseline"] or {}).get("ovl_sort") or {}
                if osrt.get("device_sorted_ovl_identical") is False:
                    parity_fail = True
                # the three stages of the reference on this box's CPUs, one after the other
                cb = out.get("cpu_baseline")
                if cb and out["overlap"]["cpu_baseline"] and "wall_s" in osrt:
                    whole_cns_s = (total_bases / args.steps / world) / cb["value"] if cb["value"] else 0.0
                    out["cpu_baseline"]["stage_chain"] = {
                        "raw_align_s": out["overlap"]["cpu_baseline"]["wall_s"], "sort_align_s": osrt["wall_s"], "seed_cns_s_extrapolated": whole_cns_s,
                        "corrected_bases_per_s": (total_bases / args.steps / world) / (out["overlap"]["cpu_baseline"]["wall_s"] + osrt["wall_s"] + whole_cns_s),
                        "note": "minimap2-nd and ovl_sort of the compiled reference timed whole on %d threads; seed_cns = this step's corrected bases / "
                                "the sample's rate (the sample is every k-th pile + the longest seeds); the metric over the three stages"
                                % out["overlap"]["cpu_baseline"]["cores"]}
                if out["overlap"]["cpu_baseline"] and out["overlap"]["cpu_baseline"].get("device_ovl_identical") is False:
                    parity_fail = True
        print(json.dumps(out))
        if parity_fail:
            db.close()
            sys.exit("bench.py: the device's records differ from the reference's (see \"parity\")")
    db.close()
    import shutil as _sh
    _sh.rmtree(out_dir, ignore_errors=True)
    if dist is not None:
        dist.barrier()
        if exchange is not None and rank == 0:   # (every rank is past its last read)
            import shutil
            shutil.rmtree(exchange.dir, ignore_errors=True)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
