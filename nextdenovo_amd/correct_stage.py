#!/usr/bin/env python
"""The whole correction stage of NextDenovo on the MI355X with nothing on disk in between.

nextDenovo runs three kinds of subtasks over the `seq_dump` output directory and exchanges files between them
(reference nextDenovo:426-467 raw_align, :344-354 sort_align, :77-84 seed_cns):

    minimap2-nd --step 1 ... input.seed.NNN.2bit input.part.MMM.2bit -o input.seed.NNN.2bit.K.ovl      (one per file pair)
    ovl_sort ... -i .input.seed.NNN.idx -o input.seed.NNN.sorted.ovl input.fofn                        (one per seed file)
    nextcorrect.py -f idxs -i input.seed.NNN.sorted.ovl -r ont ... -o cns.fasta                        (one per seed file)

This command does the same jobs in the same order on the device and hands the records from one step to the next in
memory: the step-1 overlaps never get varint-coded, sorted.ovl is never written or re-read, each seed file's minimizer
index is built once for all of its jobs, and the reads are uploaded once.

    python -m nextdenovo_amd.correct_stage -d 01.raw_align -x ava-ont -k 40 -r ont -min_len_seed 5000 -o cns
    python -m nextdenovo_amd.correct_stage --fofn input.fofn --read-cutoff 1k --seed-cutoff 10k --seed-cutfiles 2 -d 01.raw_align ...   (db_split too)

Several GPUs: start it once per GPU with torch.distributed.run (`python -m torch.distributed.run --nproc-per-node 8 --master-addr
127.0.0.1 -m nextdenovo_amd.correct_stage ...`): rank r takes the seed files r, r + N, ... (one `seed_cns` subtask per GPU, as
nextDenovo shards the stage: `seed_cutfiles = pa_correction = n_gpu`), computes every raw_align job their sorts read -- the mirrors
`(t < i, seed i)` too, so no rank waits for another -- and the ranks meet once, in an all-reduce of {corrected bases, corrected
seeds} (RCCL on a GPU node) that rank 0 prints.

writes `cns.NNN.fasta` (+ `.idx`) per seed file, byte-identical to what the three reference programs produce when the
sort's input list names the `.ovl` files in job order (the reference lists them in directory order, which only matters
for overlaps with equal (seed, match, span) keys).  `--keep DIR` also writes the `.ovl`, `sorted.ovl` and `.bl` files
with the reference's names, for resuming or debugging with the file-based commands.
"""
from __future__ import annotations

import argparse
import glob
import os
import sys

import numpy as np

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextdenovo_amd import api, minimap2_nd, nextcorrect, overlap  # noqa: E402


def job_matrix(n_seed: int, n_part: int):
    """raw_align's job list (nextDenovo:426-467): (k, seed file i, other kind, other index, dual)."""
    jobs, k = [], 0
    for i in range(n_seed):
        for j in range(n_part):
            jobs.append((k, i, "part", j, True))
            k += 1
        for t in range(i, n_seed):
            jobs.append((k, i, "seed", t, t != i))
            k += 1
    return jobs


class _IndexCache:
    """The index parts of one target file, built on first use and shared by every job that maps against it."""

    def __init__(self, opt, target):
        self.opt, self.target, self.parts, self.mid_occ = opt, target, {}, {}

    def map(self, query, batch_size, dual):
        o = overlap.Opt.from_buffer_copy(self.opt)
        o.no_dual = 0 if dual else 1
        out = []
        layout = minimap2_nd.index_parts(self.target.lens, batch_size)
        # every reference job is its own process: its occurrence threshold comes from the FIRST index part of ITS -I
        # split (main.c:489, options.c:70-71) -- kept per split layout, not per target file
        key = tuple(layout)
        for lo, hi in layout:
            ix = self.parts.get((lo, hi))
            if ix is None:
                ix = self.parts[(lo, hi)] = overlap.Index(self.opt, self.target.subset(lo, hi))
            if key not in self.mid_occ:
                self.mid_occ[key] = self.opt.mid_occ if self.opt.mid_occ > 0 else ix.mid_occ()
            out.append(ix.map(query, self.mid_occ[key], opt=o))
        return np.concatenate(out) if len(out) > 1 else out[0]

    def close(self):
        for ix in self.parts.values():
            ix.close()
        self.parts = {}


def read_db_from_sets(sets):
    """All reads of the run, indexed by read id (ids are dense over seed and part files, util/seq_dump.c:83-84)."""
    n = max(int(s.ids.max()) + 1 if len(s) else 0 for s in sets)
    word_off = np.zeros(n, dtype=np.uint64)
    lens = np.zeros(n, dtype=np.uint32)
    chunks, base = [], 0
    for s in sets:
        word_off[s.ids] = s.word_off + np.uint64(base)
        lens[s.ids] = s.lens
        chunks.append(s.words)
        base += s.words.size
    return np.concatenate(chunks), word_off, lens


def run(argv) -> int:
    ap = argparse.ArgumentParser(prog="correct_stage", description=__doc__.split("\n")[0])
    ap.add_argument("-d", dest="dir", required=True, help="seq_dump output directory (input.seed.*.2bit, input.part.*.2bit)")
    ap.add_argument("--fofn", default=None, metavar="FILE", help="start from FASTA/FASTQ[.gz] files: run seq_dump into -d first")
    ap.add_argument("--read-cutoff", default="1k", help="seq_dump -f (with --fofn)")
    ap.add_argument("--seed-cutoff", default=None, help="seq_dump -s (with --fofn)")
    ap.add_argument("--blocksize", default="0", help="seq_dump -b (with --fofn)")
    ap.add_argument("--seed-cutfiles", type=int, default=1, help="seq_dump -n (with --fofn)")
    ap.add_argument("-x", dest="preset", required=True, choices=["ava-ont", "ava-pb", "ava-hifi"])
    ap.add_argument("-f", dest="occ", default=None, help="minimap2-nd -f (FLOAT < 1 or INT)")
    ap.add_argument("-k", dest="sort_k", type=int, default=40, help="ovl_sort -k")
    ap.add_argument("-l", dest="flank", type=int, default=300, help="ovl_sort -l")
    ap.add_argument("-r", dest="read_type", required=True, type=str.lower, choices=["clr", "hifi", "ont"])
    ap.add_argument("--seed-files", default=None, help="comma separated seed file numbers to correct (default: all)")
    ap.add_argument("--keep", default=None, metavar="DIR", help="also write .ovl / sorted.ovl / .bl files there")
    ap.add_argument("-o", dest="out", required=True, help="output prefix: PREFIX.NNN.fasta (+ .idx)")
    for name, typ, dflt in (("-max_cov_aln", int, 130), ("-max_lq_length", str, None), ("-min_cov_seed", int, 10),
                            ("-min_len_seed", str, 10000), ("-min_len_aln", str, 500), ("-min_cov_base", int, 4),
                            ("-min_error_corrected_ratio", float, 0.8), ("-p", int, 0)):
        ap.add_argument(name, dest=name.lstrip("-") if name != "-p" else "process", type=typ, default=dflt)
    ap.add_argument("-s", dest="split", action="store_true")
    ap.add_argument("-fast", action="store_true")
    ap.add_argument("-b", dest="blacklist", action="store_false", default=True)
    ap.add_argument("--batch", type=int, default=4096)
    a = ap.parse_args(argv)
    if a.max_lq_length is None:
        a.max_lq_length = 10000 if a.preset == "ava-ont" else 1000  # lib/config_parser.py:217-221
    a.max_lq_length = nextcorrect.parse_num_unit(a.max_lq_length)
    a.min_len_seed = nextcorrect.parse_num_unit(a.min_len_seed)
    a.min_len_aln = nextcorrect.parse_num_unit(a.min_len_aln)
    a.read_type = {"ont": 1, "clr": 2, "hifi": 3}[a.read_type]

    if a.fofn:  # db_split (nextDenovo:536-551) first: the same files seq_dump writes, packed on the device
        from nextdenovo_amd import seq_dump
        if a.seed_cutoff is None:
            raise SystemExit("[ERROR] --fofn needs --seed-cutoff")
        if seq_dump.run(["-f", a.read_cutoff, "-s", a.seed_cutoff, "-b", a.blocksize, "-n", str(a.seed_cutfiles), "-d", a.dir, a.fofn]) != 0:
            raise SystemExit("[ERROR] seq_dump failed")
    seed_paths = sorted(glob.glob(os.path.join(a.dir, "input.seed.*.2bit")))
    part_paths = [p for p in sorted(glob.glob(os.path.join(a.dir, "input.part.*.2bit"))) if os.path.getsize(p) > 2]
    if not seed_paths:
        raise SystemExit("[ERROR] no input.seed.*.2bit in %s" % a.dir)
    seeds = [overlap.ReadSet.from_2bit(p) for p in seed_paths]
    parts = [overlap.ReadSet.from_2bit(p) for p in part_paths]
    want = set(range(len(seeds))) if a.seed_files is None else {int(x) - 1 for x in a.seed_files.split(",")}
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:  # one rank per GPU: seed file i goes to rank i mod N (nextDenovo:77-84 runs one seed_cns subtask per seed file)
        os.environ.setdefault("NDGPU_DEVICE", os.environ.get("LOCAL_RANK", "0"))
        want = {i for i in want if i % world == rank}

    argv_mm = ["--step", "1", "-x", a.preset] + (["-f", a.occ] if a.occ else []) + ["a", "b"]
    opt = minimap2_nd.build_opt(minimap2_nd.parse_argv(argv_mm))
    seed_batch = 6000000000 if a.preset == "ava-hifi" else 3000000000  # -I 6G / 3G of the seed x seed jobs (nextDenovo:430,456)

    # --- raw_align: every job once, records kept in host memory
    per_seed_files = {i: [] for i in range(len(seeds))}   # seed file -> [(k, records)] of the jobs that involve it
    caches = {}
    for k, i, kind, j, dual in job_matrix(len(seeds), len(parts)):
        if i not in want and not (kind == "seed" and j in want):
            continue
        if i not in caches:
            caches[i] = _IndexCache(opt, seeds[i])
        query = parts[j] if kind == "part" else seeds[j]
        recs = caches[i].map(query, minimap2_nd.IDX_BATCH if kind == "part" else seed_batch, dual)
        per_seed_files[i].append((k, recs))
        if kind == "seed" and j != i:
            per_seed_files[j].append((k, recs))          # the `ln -sf` mirror of a seed x seed job (nextDenovo:459)
        if a.keep:
            os.makedirs(a.keep, exist_ok=True)
            name = "%s.%d.ovl" % (os.path.basename(seed_paths[i]), k)
            with open(os.path.join(a.keep, name), "wb") as f:
                f.write(overlap.encode(recs, np.zeros(2, dtype=np.uint32)))
        if kind == "seed" and j == len(seeds) - 1:
            caches.pop(i).close()                         # seed file i is not a target again
    for c in caches.values():
        c.close()

    # --- sort_align + seed_cns per seed file
    words, word_off, lens = read_db_from_sets(seeds + parts)
    db = api.ReadDB(words, word_off, lens)
    fail = 0
    totals = [0, 0]  # corrected bases, corrected records of this rank
    # The seed files run through the stage like parts through a line (stage.StagePipeline): the sort + pile admission of the next one on a
    # thread of its own while this one's consensus holds the device, and two consensus calls in flight -- the tail of one (host ranking,
    # POA, two rounds of small launches) under the main phases of the next.  NDGPU_STAGE_SERIAL=1: one seed file after the other.
    def make_piles(i):
        s = seeds[i]
        seed_len = np.zeros(int(lens.size), dtype=np.uint32)
        seed_len[s.ids] = s.lens
        files = [r for _, r in sorted(per_seed_files[i], key=lambda kr: kr[0])]
        srt, bl, _ = overlap.sort_overlaps(files, seed_len, int(s.lens.min()) if len(s) else 0, a.sort_k, a.flank)
        tag = os.path.basename(seed_paths[i])[len("input.seed."):-len(".2bit")]
        if a.keep:
            so = os.path.join(a.keep, "input.seed.%s.sorted.ovl" % tag)
            with open(so, "wb") as f:
                f.write(overlap.encode(srt, np.zeros(2, dtype=np.uint32)))
            with open(so + ".bl", "w") as f:
                for rid, kind in bl:
                    f.write("%d %s\n" % (rid, kind))
        skip = [rid for rid, _ in bl] if a.blacklist else []
        dec, off, names = overlap.assemble_piles(srt, int(lens.size), a.min_len_seed, a.min_len_aln, a.max_cov_aln, a.min_cov_seed, skip)
        return tag, dec, off, names

    def correct(i, made):
        tag, dec, off, names = made
        piles = [(int(names[p]), np.arange(int(off[p]), int(off[p + 1]))) for p in range(names.size)]
        out = "%s.%s.fasta" % (a.out, tag)
        with open(out, "w") as OUT, open(out + ".idx", "w") as IDX:
            failed = nextcorrect.correct_and_write(db, dec, piles, a, OUT, IDX)
        n_bases = n_recs = 0
        for line in open(out + ".idx"):
            ln = int(line.split("\t")[2])
            if ln > 0:
                n_bases += ln
                n_recs += 1
        return failed, n_bases, n_recs

    try:
        from .stage import StagePipeline
        serial = bool(os.environ.get("NDGPU_STAGE_SERIAL"))
        for _i, (failed, n_bases, n_recs), _t in StagePipeline(make_piles, correct, depth=1 if serial else 2, prefetch=not serial).run(sorted(want)):
            fail += failed
            totals[0] += n_bases
            totals[1] += n_recs
    finally:
        db.close()
    if world > 1:  # the stage's only collective: the final counts (RCCL over xGMI on a GPU node, gloo without one)
        import torch
        import torch.distributed as dist
        use_gpu = torch.cuda.is_available()
        if use_gpu:
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl" if use_gpu else "gloo")
        t = torch.tensor(totals, dtype=torch.int64, device="cuda" if use_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        totals = [int(x) for x in t.tolist()]
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stderr.write("[correct_stage] %d corrected seeds, %d corrected bases over %d rank%s\n"
                         % (totals[1], totals[0], world, "" if world == 1 else "s"))
    return 1 if fail > 5 else 0


if __name__ == "__main__":
    sys.exit(run(sys.argv[1:]))
