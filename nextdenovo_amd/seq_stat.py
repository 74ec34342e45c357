#!/usr/bin/env python
"""`seq_stat`: read-length statistics and the suggested seed cut-off of the db_stat task (reference nextDenovo:553-563,
util/seq_stat.c) -- the same report, byte for byte.

    python -m nextdenovo_amd.seq_stat -f 1k -g 5m -d 45 [-a] [-o input.reads.stat] input.fofn

Host arithmetic only (a histogram over read lengths); it shares the kseq-faithful parser of `seq_dump`.
"""
from __future__ import annotations

import os
import sys

import numpy as np

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextdenovo_amd.seq_dump import parse_num, read_records  # noqa: E402

MIN_SEED_CUTOFF = 10000  # util/seq_stat.c:11-12
MIN_SEED_DEPTH = 20


def report(lengths, filter_len, filter_bases, filter_length, genome_size, depth, adjust) -> str:
    """out_stat (util/seq_stat.c:56-148) on the kept read lengths."""
    L = sorted((int(x) for x in lengths), reverse=True)
    n = len(L)
    total = sum(L)
    out = []
    bin_ = n // 1000 or 10
    step = 1000
    out.append("[Read length histogram ('*' =~ %d reads)]" % bin_)
    bin_count, bin_index = 0, 1
    count, length = [0] * 12, [0] * 12
    ns_index, ns_bases = 1, 0
    seed_cutoff = 0
    seed_depth = float(depth)
    seed_cov_len = depth * genome_size
    for i in range(n):
        if seed_cutoff == 0:
            seed_cov_len -= L[i]
            if seed_cov_len <= L[i]:
                seed_cutoff = L[i]
        v = L[n - 1 - i]
        if v < step * bin_index or (n - 1 - i) < (bin_ << 1):
            bin_count += 1
        else:
            while True:
                out.append("\n%7d %7d %10d  " % (step * (bin_index - 1), step * bin_index - 1, bin_count))
                out.append("*" * (bin_count // bin_))
                bin_count = 0
                bin_index += 1
                if not v > step * bin_index:
                    break
            bin_count = 1
        ns_bases += L[i]
        count[ns_index - 1] += 1
        if ns_bases >= ns_index * 0.1 * total:
            length[ns_index - 1] = L[i]
            count[ns_index] += count[ns_index - 1]
            ns_index += 1
    if adjust and seed_cutoff < MIN_SEED_CUTOFF:    # recal_seed_cutoff (:41-53)
        i, cov = 0, 0
        while i < n and L[i] >= MIN_SEED_CUTOFF:
            cov += L[i]
            i += 1
        if cov // genome_size < MIN_SEED_DEPTH:
            while i < n and cov < genome_size * (MIN_SEED_DEPTH + 5):
                cov += L[i]
                i += 1
        seed_depth = cov / genome_size
        seed_cutoff = L[i - 1]
    elif not seed_cutoff:
        seed_depth = total / genome_size
        seed_cutoff = filter_length
    if seed_cutoff == filter_length:
        seed_cutoff += 1
    out.append("\n%7d %7d %10d  " % (step * (bin_index - 1), L[0], bin_count))
    out.append("*" * (bin_count // bin_))
    out.append("\n\n[Read length stat]\n")
    out.append("%5s %20s %10s\n" % ("Types", "Count (#)", "Length (bp)"))
    for i in range(9):
        out.append("N%-4d %20d %7d\n" % ((i + 1) * 10, count[i], length[i]))
    g32 = np.float32(genome_size)

    def f32(x):  # `uint64 / (float) genome_size`: the division is done in single precision (:136-139)
        return float(np.float32(x) / g32)

    out.append("\n%-8s %20s %20s %10s\n" % ("Types", "Count (#)", "Bases (bp)", "Depth (X)"))
    out.append("%-8s %20d %20d %10.2f\n" % ("Raw", n + filter_len, total + filter_bases, f32(total + filter_bases)))
    out.append("%-8s %20d %20d %10.2f\n" % ("Filtered", filter_len, filter_bases, f32(filter_bases)))
    out.append("%-8s %20d %20d %10.2f\n" % ("Clean", n, total, f32(total)))
    out.append("\n*Suggested seed_cutoff (genome size: %.2fMb, expected seed depth: %d, real seed depth: %.2f): %d bp\n"
               % (genome_size / 1000000, depth, seed_depth, seed_cutoff))
    if seed_cutoff < MIN_SEED_CUTOFF:
        out.append("\033[35m*NOTE:\033[0m The read/seed length is too short, and the assembly result is unexpected and please check"
                   " the assembly quality carefully. Of course, it's better to sequencing more longer reads and try again.\n")
    return "".join(out)


def run(argv) -> int:
    import getopt
    opts, args = getopt.getopt(argv, "f:g:d:o:a")
    o = dict(opts)
    if len(args) < 1:
        sys.stderr.write("Usage: seq_stat [options] input.fofn\n")
        return 1
    filter_length = parse_num(o.get("-f", "1000"))
    genome_size = parse_num(o.get("-g", "5000000"))
    depth = parse_num(o.get("-d", "45"))
    adjust = "-a" not in o
    fofn = args[0]
    base = os.path.dirname(fofn) or "."
    lengths, filter_len, filter_bases = [], 0, 0
    with open(fofn) as f:
        lines = f.read().split("\n")
    for line in lines:
        if len(line) == 0 or line.startswith("#"):
            continue
        path = line if line.startswith("/") else os.path.join(base, line)
        _, recs = read_records(path)
        for _, l in recs:
            if l < filter_length:
                filter_len += 1
                filter_bases += l
            else:
                lengths.append(l)
    if not lengths:
        if "-o" in o:
            open(o["-o"], "wb").close()
        return 0
    text = report(lengths, filter_len, filter_bases, filter_length, genome_size, depth, adjust)
    if "-o" in o:
        with open(o["-o"], "w") as f:
            f.write(text)
    else:
        sys.stdout.write(text)
    return 0


if __name__ == "__main__":
    sys.exit(run(sys.argv[1:]))
