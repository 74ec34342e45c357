#!/usr/bin/env python
"""`minimap2-nd --step 1` and `--step 2` on the MI355X: all-vs-all read overlap, `.ovl` out.

Takes the command line nextDenovo writes for the raw-align subtasks (reference nextDenovo:436-466):

    python -m nextdenovo_amd.minimap2_nd --step 1 [--dual=yes] [--mode 3] -t 8 -x ava-ont|ava-pb|ava-hifi [-f N[,M]] [-I 4G] target.2bit query.2bit -o out.ovl

and writes the byte-identical overlap file.  Mirrors minimap2/main.c for the options of this path (preset
first, then the remaining options in order: main.c:140-366); the index is split into parts exactly as
mm_idx_gen does with -I (index.c:284-287,351-360), the occurrence threshold comes from the first part
(options.c:70-71), every query file is mapped against every part in turn (main.c:474-507).

`--step 2` runs as nextDenovo runs it (--mode 2, the default: every marked candidate mapped again with the short k-mer sketch) or
with `--mode 0`.  `--mode 3` (HiFi: chain ends trimmed, every hit extended into the unaligned read ends, minimap2/map.c:340-482) is built in.
`--step 2` is the `cns_align` command of nextDenovo:356-366 on corrected reads: FASTA input with numeric names; the hits of every
read on the device, the per-target marking, the re-alignment's bookkeeping (its mapping passes run on the device again), the record
filters, the dovetail / contained filter, the 10-field encoder and the `.bl` table on the host (csrc/ovl_step2.cpp).  `--mode 1` too
(the two forms of the re-alignment switch at 20 candidates instead of 200, --cn 50; its one-read-index mappings chain through
mm_chain_dp_nextdenovo, anchor thinning beyond 100,000 anchors included).
`-c` (with --step 1, not --mode 3, not ava-hifi -- the compiled reference aborts there): base-level alignment through every chain
(mm_align_skeleton, minimap2/align.c:857-913) before the writer's filter; -A -B -O -E -z -s as in minimap2/main.c:250-252,353-361.
`-f FLOAT,INT` re-chains the reads that found no chain with the second threshold (map.c:553-575); `-n` may go down to 1; `-k` up to
127 (not 32, 64, 96: the reference's own mask is undefined there).
Options of other paths (-a, --step 3) are rejected, not approximated.
"""
from __future__ import annotations

import sys

import numpy as np

from . import overlap

IDX_MINI_BATCH = 50000000      # mm_idxopt_init (options.c:9)
IDX_BATCH = 4000000000         # options.c:10


def parse_num(s: str) -> int:
    """mm_parse_num (minimap2/misc / main.c): float with optional K/M/G suffix."""
    mult = 1.0
    if s and s[-1] in "gG":
        mult, s = 1e9, s[:-1]
    elif s and s[-1] in "mM":
        mult, s = 1e6, s[:-1]
    elif s and s[-1] in "kK":
        mult, s = 1e3, s[:-1]
    return int(float(s) * mult + .499)


def yes_no(v: str) -> bool:
    return v.lower() in ("yes", "y")


class Args:
    def __init__(self):
        self.preset = None
        self.step = 0
        self.out = None
        self.batch_size = IDX_BATCH
        self.kn, self.wn, self.cn = 17, 10, 20   # --step 2 (main.c:197)
        self.cigar = False        # -c
        self.aopt = None          # the scoring options of -c (overlap.AlnOpt)
        self.files = []
        self.ops = []  # (name, value) in command-line order, applied after the preset


LONG_WITH_ARG = {"--step", "--minlen", "--maxhan1", "--maxhan2", "--seed", "--dual", "--mode", "--df", "--minide", "--minmatch", "--kn", "--wn", "--cn", "--cap-sw-mem"}
SHORT_WITH_ARG = set("xtfIKkwornmgsNpMABOEz")


def parse_argv(argv) -> Args:
    a = Args()
    i = 0
    while i < len(argv):
        tok = argv[i]
        if tok.startswith("--"):
            name, _, val = tok.partition("=")
            if name in LONG_WITH_ARG and not _:
                i += 1
                val = argv[i]
            a.ops.append((name, val))
        elif tok.startswith("-") and len(tok) > 1:
            c = tok[1]
            if c in SHORT_WITH_ARG:
                val = tok[2:]
                if not val:
                    i += 1
                    val = argv[i]
                a.ops.append(("-" + c, val))
            else:
                for ch in tok[1:]:
                    a.ops.append(("-" + ch, None))
        else:
            a.files.append(tok)
        i += 1
    return a


def build_opt(a: Args) -> overlap.Opt:
    # first pass, as minimap2/main.c:185-200 does: -x and --step before everything else, so that --step's minlen = 500
    # default never overrides an explicit --minlen, wherever that appears on the command line
    for name, val in a.ops:
        if name == "-x":
            a.preset = val
    opt = overlap.preset(a.preset)  # raises for unsupported presets
    a.aopt = overlap.aln_opt()
    for name, val in a.ops:
        if name == "--step":
            a.step = int(val)
            if a.step == 1:
                opt.minlen = 500
            elif a.step == 2:  # main.c:194-197
                opt.step, opt.minide, opt.minlen, opt.maxhan1, opt.maxhan2, opt.minmatch = 2, 0.05, 2000, 5000, 500, 100
            else:
                raise SystemExit("[ERROR] only --step 1 and --step 2 are built in this engine")
    for name, val in a.ops:
        if name in ("-x", "-t", "--step"):
            continue
        elif name == "--dual":
            opt.no_dual = 0 if yes_no(val) else 1
        elif name == "-X":
            opt.no_diag, opt.no_dual = 1, 1
        elif name == "-f":
            head, _, tail = val.partition(",")
            x = float(head)
            if x < 1.0:
                opt.mid_occ_frac, opt.mid_occ = x, 0
            else:
                opt.mid_occ = int(x + .499)
            if tail:   # main.c:343: a read that chained nothing is seeded again up to this many occurrences (map.c:553-575)
                opt.max_occ = int(float(tail) + .499)
        elif name == "-I":
            a.batch_size = parse_num(val)
        elif name == "-K":
            pass  # query mini-batch size: affects only when the reference flushes its buffer
        elif name == "-k":
            opt.k = int(val)
        elif name == "-w":
            opt.w = int(val)
        elif name == "-H":
            opt.hpc = 1
        elif name == "-r":
            opt.bw = parse_num(val)
        elif name == "-n":
            opt.min_cnt = int(val)
        elif name == "-m":
            opt.min_chain_score = int(val)
        elif name == "-g":
            opt.max_gap = parse_num(val)
        elif name == "--minlen":
            opt.minlen = parse_num(val)
        elif name == "--maxhan1":
            opt.maxhan1 = parse_num(val)
        elif name == "--maxhan2":
            opt.maxhan2 = parse_num(val)
        elif name == "--dvt":
            opt.dvt = 1
        elif name == "--seed":
            opt.seed = int(val)
        elif name == "--mode":
            opt.mode = int(val)  # --step 1 only asks whether it is 3 (minimap2/map.c:488,919)
        elif name == "--df":
            opt.d_factor = float(val)
        elif name == "--minide":
            opt.minide = float(val)
        elif name == "--minmatch":
            opt.minmatch = parse_num(val)
        elif name == "--kn":
            a.kn = int(val)   # the re-alignment's short k-mer sketch (main.c:197,219-221)
        elif name == "--wn":
            a.wn = int(val)
        elif name == "--cn":
            a.cn = int(val)
        elif name == "-o":
            a.out = val
        elif name == "-c":
            a.cigar = True
        elif name == "-A":
            a.aopt.a = int(val)
        elif name == "-B":
            a.aopt.b = int(val)
        elif name == "-s":
            a.aopt.min_dp_max = int(val)
        elif name == "--cap-sw-mem":
            a.aopt.max_sw_mat = parse_num(val)   # main.c:305
        elif name in ("-O", "-E", "-z"):  # one value sets both (main.c:353-361)
            head, _, tail = val.partition(",")
            first, second = int(head), int(tail) if tail else int(head)
            if name == "-O":
                a.aopt.q, a.aopt.q2 = first, second
            elif name == "-E":
                a.aopt.e, a.aopt.e2 = first, second
            else:
                a.aopt.zdrop, a.aopt.zdrop_inv = first, second
        else:
            raise SystemExit("[ERROR] option %s is outside the --step 1 overlap path of this engine" % name)
    if a.step not in (1, 2):
        raise SystemExit("[ERROR] --step 1 or --step 2 is required")
    if a.step == 2 and opt.mode not in (0, 1, 2):
        raise SystemExit("[ERROR] --step 2 runs with --mode 2 (the default: every marked candidate mapped again), 1 or 0 (no re-alignment); "
                         "--mode 3 belongs to --step 1")
    if a.step == 2 and opt.mode == 1:   # main.c:455-457
        opt.minide = max(opt.minide, 0.01)
        if a.cn == 20:
            a.cn = 50
    if a.cigar and (a.step != 1 or opt.mode == 3):
        raise SystemExit("[ERROR] -c is built for --step 1 without --mode 3")
    if a.step == 2 and not a.out:
        raise SystemExit("[ERROR] --step 2 needs -o FILE (the .bl table is written next to it, main.c:262-272)")
    return opt


def index_parts(lens: np.ndarray, batch_size: int, mini_batch=IDX_MINI_BATCH):
    """Read ranges of the index parts, as mm_idx_gen forms them: mini-batches of >= min(mini_batch, batch_size)
    bases are appended while the running total is still <= batch_size."""
    mbs = min(mini_batch, batch_size)
    n, i, parts = int(lens.size), 0, []
    while i < n:
        start, total = i, 0
        while i < n and total <= batch_size:
            size = 0
            while i < n:
                size += int(lens[i])
                i += 1
                if size >= mbs:
                    break
            total += size
        parts.append((start, i))
    return parts


def load_reads(path: str) -> overlap.ReadSet:
    """`.2bit` as seq_dump writes it, or FASTA / FASTQ[.gz] whose names are numbers (the corrected-read files `cns.fasta` that
    --step 2 maps: minimap2-nd takes strtoul of the name, minimap2/map.c:1298-1300), packed on the device."""
    if path.endswith(".2bit"):
        return overlap.ReadSet.from_2bit(path)
    from . import seq_dump
    bufs, offs, lens, ids, base = [], [], [], [], 0
    for b, off, ln, nm in seq_dump.iter_chunks(path, names=True):
        bufs.append(b), offs.append(off + np.uint64(base)), lens.append(ln), ids.append(nm)
        base += b.size
    if not bufs:
        return overlap.ReadSet(np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(1, np.uint32), np.zeros(0, np.uint64))
    buf, off, ln, nm = np.concatenate(bufs), np.concatenate(offs), np.concatenate(lens), np.concatenate(ids)
    words, word_off = overlap.pack_2bit(buf, off, ln)
    return overlap.ReadSet(nm, ln, words, word_off)


def run(argv) -> int:
    a = parse_argv(argv)
    opt = build_opt(a)
    if len(a.files) < 2:
        raise SystemExit("[ERROR] missing input: target query [query ...]")
    target = load_reads(a.files[0])
    queries = [load_reads(f) if f != a.files[0] else target for f in a.files[1:]]
    prev = np.zeros(2, dtype=np.uint32)  # `prev_t pid` lives for the whole run (main.c:29)
    mid_occ = opt.mid_occ
    out = open(a.out, "wb") if a.out else sys.stdout.buffer
    flt = overlap.Step2Filter() if a.step == 2 else None
    try:
        if flt:
            out.write(b"\x00\xff")  # init_ovl_mode(stdout, 10), lib/ovl.c:70-75
        realign = a.step == 2 and opt.mode in (1, 2)
        mini_opt, q_minis = None, {}
        if realign:  # the re-alignment's indexes use the short k-mer sketch (--kn 17 --wn 10, main.c:197), hpc as the preset's
            mini_opt = overlap.Opt.from_buffer_copy(opt)
            mini_opt.k, mini_opt.w = a.kn, a.wn
        for lo, hi in index_parts(target.lens, a.batch_size):
            part = target.subset(lo, hi)
            with overlap.Index(opt, part) as ix:
                if mid_occ <= 0:
                    mid_occ = ix.mid_occ()
                t_mini = overlap.Index(mini_opt, part) if realign else None
                try:
                    for qi, q in enumerate(queries):
                        if realign:
                            whole = q is target and lo == 0 and hi == len(target)   # the same reads: one short-sketch index serves both sides
                            if qi not in q_minis:
                                q_minis[qi] = t_mini if whole else overlap.Index(mini_opt, q)
                            out.write(flt.feed(ix.map2_realign(part, q, mid_occ, q_minis[qi], t_mini, a.cn), opt.maxhan1, opt.maxhan2))
                            if whole:
                                del q_minis[qi]   # (it is closed with the part)
                        elif flt:
                            out.write(flt.feed(ix.map2(q, mid_occ), opt.maxhan1, opt.maxhan2))
                        elif a.cigar:
                            out.write(overlap.encode(ix.map_cigar(part, q, mid_occ, a.aopt), prev))
                        else:
                            out.write(overlap.encode(ix.map(q, mid_occ), prev))
                finally:
                    if t_mini is not None:
                        t_mini.close()
        for m in q_minis.values():
            m.close()
        if flt:
            with open(a.out + ".bl", "w") as f:
                f.write(flt.bl())
    finally:
        if flt:
            flt.close()
        if a.out:
            out.close()
    return 0


if __name__ == "__main__":
    sys.exit(run(sys.argv[1:]))
