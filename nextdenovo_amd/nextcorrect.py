#!/usr/bin/env python
"""Stage CLI of the MI355X engine: same command line, inputs and outputs as the reference's
`lib/nextcorrect.py` (argument parser :272-331, pile assembly :92-143, output :233-260), but the
piles of a whole sorted.ovl are corrected in batches on the GPU against a read DB that is
uploaded once (api.ReadDB) instead of one ctypes call per pile in a fork pool.

    python -m nextdenovo_amd.nextcorrect -f idxs.fofn -i input.seed.001.sorted.ovl -r ont -o cns.fasta

Output records come out in sorted.ovl seed order, i.e. exactly the reference at `-p 1`
(its `imap_unordered` makes the order nondeterministic for -p > 1, lib/nextcorrect.py:233).
`-p` sets the number of host threads; `-dbuf` is accepted and ignored (the DB lives in HBM).
"""
from __future__ import annotations

import argparse
import os
import re
import sys

import numpy as np

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextdenovo_amd import api, ovl  # noqa: E402


def parse_num_unit(v):
    """lib/kit.py parse_num_unit: 10k / 5m / 1g suffixes."""
    s = str(v).strip().lower()
    mult = {"k": 10 ** 3, "m": 10 ** 6, "g": 10 ** 9}
    if s and s[-1] in mult:
        return int(float(s[:-1]) * mult[s[-1]])
    return int(float(s))


def read_blacklist(path, skip):
    if path and os.path.exists(str(path)):
        with open(path) as f:
            for line in f:
                if line.strip():
                    skip.add(int(line.strip().split()[0]))


def read_corrected_seeds(IN, skip):
    """Resume support, lib/nextcorrect.py:156-181: drop the last (possibly partial) idx record."""
    i, offset, seed_name, last_line = 0, -1, 0, ""
    last_valid = [0, 0, -1]
    line = IN.readline()
    while line:
        offset = IN.tell()
        seed_name = re.split(r"[_,\s]+", line.strip().split()[0])[0]
        skip.add(int(seed_name))
        if last_line:
            last_valid = last_line.strip().split()
        last_line = line
        i += 1
        line = IN.readline()
    if offset != -1:
        IN.seek(offset - len(last_line), 0)
        IN.truncate()
        skip.discard(int(seed_name))
    return int(last_valid[1]) + int(last_valid[2]) + 1


def assemble_piles(recs: np.ndarray, args, skip):
    """lib/nextcorrect.py:92-143 read_seq_data, on the decoded record array.
    Yields (seed, record index list)."""
    count = total_length = seed_length = 0
    used, seed_name, rows = set(), "", []
    last_seed = -1
    lim = args.max_cov_aln * 1.5
    for k in range(recs.shape[0]):
        t_name, _, t_s, t_e, q_name, q_s, q_e, match = (int(x) for x in recs[k])
        if seed_name == "+" or (last_seed != -1 and t_name != last_seed):
            if seed_length and total_length / seed_length >= args.min_cov_seed and seed_name != "+":
                yield seed_name, rows
            used, seed_name, rows = set(), "", []
            total_length = seed_length = 0
        if seed_name == "":
            seed_length = t_e + 1
            total_length = 0
            seed_name = t_name if seed_length >= args.min_len_seed and t_name not in skip else "+"
        if t_e - t_s < args.min_len_aln or total_length / seed_length > lim or q_name in used or seed_name == "+":
            continue
        rows.append(k)
        used.add(q_name)
        total_length += t_e - t_s + 1
        last_seed = t_name
    if seed_length and total_length / seed_length >= args.min_cov_seed and seed_name not in ("+", ""):
        yield seed_name, rows


def assemble_piles_fast(recs: np.ndarray, min_len_seed: int, min_len_aln: int, max_cov_aln: int, min_cov_seed: int, skip=()):
    """Vectorised form of assemble_piles (same admission rules, lib/nextcorrect.py:92-143) for large record arrays.
    Returns (rows, pile_off, seeds): rows = indices into recs of the admitted records of every valid pile, in order."""
    n = recs.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(1, dtype=np.uint64), np.zeros(0, dtype=np.uint32)
    t = recs[:, 0].astype(np.int64)
    q = recs[:, 4].astype(np.int64)
    span = recs[:, 3].astype(np.int64) - recs[:, 2].astype(np.int64)
    first = np.flatnonzero(np.r_[True, t[1:] != t[:-1]])          # first record of every seed group
    gid = np.cumsum(np.r_[True, t[1:] != t[:-1]]) - 1
    seed_len = recs[first, 3].astype(np.int64) + 1                 # the group's first record is the self record
    seed_ok = seed_len >= min_len_seed
    if len(skip):
        seed_ok &= ~np.isin(t[first], np.fromiter(skip, dtype=np.int64, count=len(skip)))
    ok = (span >= min_len_aln) & seed_ok[gid]
    # one overlap per query read: first occurrence among the length-admitted records of the group
    idx = np.flatnonzero(ok)
    order = np.lexsort((idx, q[idx], gid[idx]))
    so = idx[order]
    dup = np.r_[False, (gid[so][1:] == gid[so][:-1]) & (q[so][1:] == q[so][:-1])]
    ok[so[dup]] = False
    # cumulative depth limit: total_length before the record over seed length must not exceed 1.5 x max_cov_aln
    add = np.where(ok, span + 1, 0)
    cum = np.cumsum(add) - add
    cum -= cum[first][gid]
    ok &= cum / seed_len[gid] <= max_cov_aln * 1.5
    add = np.where(ok, span + 1, 0)
    total = np.add.reduceat(add, first)
    valid = seed_ok & (total / seed_len >= min_cov_seed)
    ok &= valid[gid]
    rows = np.flatnonzero(ok)
    counts = np.add.reduceat(ok.astype(np.int64), first)[valid]
    pile_off = np.zeros(counts.size + 1, dtype=np.uint64)
    np.cumsum(counts, out=pile_off[1:])
    return rows, pile_off, t[first][valid].astype(np.uint32)


def correct_and_write(db, recs, piles, args, OUT, IDX):
    """Correct `piles` = [(seed, row indices into recs)] in device batches and write the records as
    lib/nextcorrect.py:233-260 does.  Returns the number of seeds that failed with len == 3."""
    corrected_region = re.compile(r"[ACGT]+")
    fail_seed = 0
    batch = max(1, args.batch)
    for b0 in range(0, len(piles), batch):
        chunk = piles[b0:b0 + batch]
        sub = np.ascontiguousarray(np.concatenate([recs[rows] for _, rows in chunk]))
        off = np.zeros(len(chunk) + 1, dtype=np.uint64)
        np.cumsum([len(rows) for _, rows in chunk], out=off[1:])
        res = db.correct_piles(sub, off, min_len_aln=args.min_len_aln, max_cov_aln=args.max_cov_aln,
                               min_cov_base=args.min_cov_base, max_lq_length=args.max_lq_length,
                               min_error_corrected_ratio=args.min_error_corrected_ratio, split=int(args.split),
                               fast=int(args.fast), read_type=args.read_type, host_threads=args.process)
        for (seed_name, _), (ln, identity, seq) in zip(chunk, res):
            seq = seq.decode()
            if ln >= args.min_len_seed and identity >= args.min_error_corrected_ratio:
                if args.split:
                    regions = corrected_region.findall(seq)
                    for i, reg in enumerate(regions):
                        if len(reg) >= args.min_len_seed:
                            print(">%s_%d %d %f\n%s" % (seed_name, i + 1, len(reg), 1, reg), file=OUT)
                            if IDX:
                                print("%s_%d\t%d\t%d" % (seed_name, i + 1, OUT.tell() - len(reg) - 1, len(reg)), file=IDX)
                else:
                    print(">%s %d %f\n%s" % (seed_name, ln, identity, seq), file=OUT)
                    if IDX:
                        print("%d\t%d\t%d" % (seed_name, OUT.tell() - ln - 1, ln), file=IDX)
            else:
                if ln == 3:
                    fail_seed += 1
                elif IDX:
                    print("%d\t%d\t%d" % (seed_name, 0, 0), file=IDX)
    return fail_seed


def main(args):
    OUT, IDX = sys.stdout, None
    skip = set()
    read_blacklist(args.blacklist, skip)
    if args.out != "stdout":
        if os.path.exists(args.out):
            IDX = open(args.out + ".idx", "r+")
            pos = read_corrected_seeds(IDX, skip)
            OUT = open(args.out, "r+")
            OUT.seek(pos, 0)
            OUT.truncate()
        else:
            OUT = open(args.out, "w")
            IDX = open(args.out + ".idx", "w")

    words, word_off, lens = ovl.load_read_db(args.idxs)
    db = api.ReadDB(words, word_off, lens)
    recs = ovl.decode_ovl(args.ovl)
    # pile admission (lib/nextcorrect.py:92-143): one native pass (ndgpu_assemble_piles) instead of a Python loop over every
    # record; `assemble_piles` above is the line-by-line restatement the tests hold it against
    from nextdenovo_amd import overlap
    n_ids = max(int(lens.size), int(recs[:, [0, 4]].max()) + 1 if recs.size else 0)
    recs, off, names = overlap.assemble_piles(overlap.from_decoded(recs), n_ids, args.min_len_seed, args.min_len_aln, args.max_cov_aln,
                                              args.min_cov_seed, sorted(skip))
    piles = [(int(names[p]), np.arange(int(off[p]), int(off[p + 1]))) for p in range(names.size)]

    fail_seed = correct_and_write(db, recs, piles, args, OUT, IDX)
    db.close()
    if args.out != "stdout":
        OUT.close()
        IDX.close()
    if fail_seed > 5:
        sys.exit(1)


def build_parser():
    p = argparse.ArgumentParser(description="correct seed reads on an MI355X (drop-in for lib/nextcorrect.py)")
    p.add_argument("-f", "--idxs", metavar="FILE", required=True)
    p.add_argument("-i", "--ovl", metavar="FILE", required=True)
    p.add_argument("-r", "--read_type", required=True, type=str.lower, choices=["clr", "hifi", "ont"])
    p.add_argument("-b", "--blacklist", action="store_false", default=True)
    p.add_argument("-o", "--out", metavar="FILE", default="stdout")
    p.add_argument("-p", "--process", type=int, default=0, help="host threads (0 = all)")
    p.add_argument("-s", "--split", action="store_true", default=False)
    p.add_argument("-dbuf", action="store_true", default=False)
    p.add_argument("-fast", action="store_true", default=False)
    p.add_argument("-max_cov_aln", type=int, default=130)
    p.add_argument("-max_lq_length", type=str, default=10000)
    p.add_argument("-min_cov_seed", type=int, default=10)
    p.add_argument("-min_len_seed", type=str, default=10000)
    p.add_argument("-min_len_aln", type=str, default=500)
    p.add_argument("-min_cov_base", type=int, default=4)
    p.add_argument("-min_error_corrected_ratio", type=float, default=0.8)
    p.add_argument("-debug", action="store_true", default=False)
    p.add_argument("--batch", type=int, default=4096, help="piles per device batch (additive option)")
    return p


def cli(argv=None):
    args = build_parser().parse_args(argv)
    args.max_lq_length = parse_num_unit(args.max_lq_length)
    args.min_len_seed = parse_num_unit(args.min_len_seed)
    args.min_len_aln = parse_num_unit(args.min_len_aln)
    args.read_type = {"ont": 1, "clr": 2, "hifi": 3}[args.read_type]
    args.blacklist = args.ovl + ".bl" if args.blacklist else None
    main(args)


if __name__ == "__main__":
    cli()
