#!/usr/bin/env python
"""`seq_dump` with the 2-bit packing on the MI355X: FASTA/FASTQ[.gz] reads -> `input.{seed,part}.NNN.2bit` + hidden `.idx`.

Takes the command line nextDenovo writes for the db_split task (reference nextDenovo:536-551, util/seq_dump.c:168-252):

    python -m nextdenovo_amd.seq_dump -f 1k -s 10k -b 2g -n 2 -d 01.raw_align input.fofn

and writes byte-identical files: reads shorter than -f are dropped, reads in [-f, -s) go to part files (a new file
whenever the running length exceeds -b), reads of at least -s (and shorter than 1,000,000) are dealt round-robin to the
-n seed files, ids are assigned in input order over both kinds (util/seq_dump.c:74-114); every read is stored as
`u32 id, u32 len, ceil(len/16) u32` with 16 bases per word, first base in the top bits (lib/bseq.c:114-139), and indexed
as `id \\t offset + 8 \\t len` (util/seq_dump.c:36-41).  Parsing follows kseq.h (multi-line FASTA / FASTQ, `\\r\\n`,
a FASTQ record whose quality length differs from its sequence length ends the file) and is native: `ndgpu_fastx_*`
(csrc/fastx_reader.cpp) streams the file through a 1 MB inflate window and hands the reads out in chunks of at most 1 Gb, so a
multi-GB `.fastq.gz` never sits in memory; every chunk is packed by one device launch (`ndgpu_pack_2bit`); there is no CPU
packing path.  (`read_records_py` is the same parser in Python, kept as the cross-check of the native one in the tests.)
"""
from __future__ import annotations

import gzip
import os
import sys

import numpy as np

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextdenovo_amd import overlap  # noqa: E402

LEN_LIMIT = 1000000  # util/seq_dump.c:13


def parse_num(s: str) -> int:
    """mm_parse_num of util/seq_dump.c:150-159 (strtod + K/M/G suffix)."""
    import re
    m = re.match(r"\s*([-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?))", s)
    x = float(m.group(1)) if m else 0.0
    rest = s[m.end():] if m else s
    if rest[:1] in ("G", "g"):
        x *= 1e9
    elif rest[:1] in ("M", "m"):
        x *= 1e6
    elif rest[:1] in ("K", "k"):
        x *= 1e3
    return int(x + .499)


CHUNK_BASES = 1 << 30   # bases handed to one packing launch
CHUNK_RECS = 1 << 20


def iter_chunks(path: str, chunk_bases: int = CHUNK_BASES, chunk_recs: int = CHUNK_RECS, names: bool = False):
    """kseq_read over one file through the native streaming reader: yields (buffer uint8, offsets uint64, lengths uint32) per
    chunk -- the sequences copied back to back, line breaks removed; with names=True also strtoul(name) of every record."""
    import ctypes as C
    lib = overlap.load()
    lib.ndgpu_fastx_open.argtypes = [C.c_char_p]
    lib.ndgpu_fastx_open.restype = C.c_void_p
    lib.ndgpu_fastx_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int64]
    lib.ndgpu_fastx_read.restype = C.c_int64
    lib.ndgpu_fastx_read_named.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.ndgpu_fastx_read_named.restype = C.c_int64
    lib.ndgpu_fastx_pending.argtypes = [C.c_void_p]
    lib.ndgpu_fastx_pending.restype = C.c_uint64
    lib.ndgpu_fastx_close.argtypes = [C.c_void_p]
    h = lib.ndgpu_fastx_open(os.fsencode(path))
    if not h:
        raise OSError("Error! %s does not exist!" % path)
    try:
        buf = np.empty(chunk_bases, dtype=np.uint8)
        off = np.empty(chunk_recs, dtype=np.uint64)
        ln = np.empty(chunk_recs, dtype=np.uint32)
        ids = np.empty(chunk_recs, dtype=np.uint32) if names else None
        while True:
            n = lib.ndgpu_fastx_read_named(h, buf.ctypes.data, buf.size, off.ctypes.data, ln.ctypes.data, ids.ctypes.data if names else None,
                                           chunk_recs)
            if n == -4:  # one read longer than the chunk: make room for it
                buf = np.empty(int(lib.ndgpu_fastx_pending(h)) + 16, dtype=np.uint8)
                continue
            if n < 0:
                raise OSError("read error in %s" % path)
            if n == 0:
                return
            used = int(off[n - 1]) + int(ln[n - 1])
            if names:
                yield buf[:used].copy(), off[:n].copy(), ln[:n].copy(), ids[:n].copy()
            else:
                yield buf[:used].copy(), off[:n].copy(), ln[:n].copy()
    finally:
        lib.ndgpu_fastx_close(h)


def iter_chunks_files(paths, threads: int = 0, depth: int = 4, **kw):
    """iter_chunks over several files IN ORDER, the files being read (inflated, parsed) by up to `threads` reader threads at the same
    time -- one gzip stream cannot be split, but input.fofn usually lists many files and the reference reads them one after the
    other (util/seq_dump.c:60-72).  The native reader releases the interpreter lock, so the threads really run side by side; every
    file's chunks pass through a bounded queue (`depth` chunks), the consumer sees exactly what the sequential loop would yield."""
    import queue
    import threading
    paths = list(paths)
    threads = threads or min(len(paths), os.cpu_count() or 1, 8)
    if threads <= 1 or len(paths) <= 1:
        for p in paths:
            yield from iter_chunks(p, **kw)
        return
    queues = [queue.Queue(maxsize=depth) for _ in paths]
    stop = threading.Event()

    def reader(i):
        try:
            for chunk in iter_chunks(paths[i], **kw):
                while not stop.is_set():
                    try:
                        queues[i].put(chunk, timeout=0.2)
                        break
                    except queue.Full:
                        pass
                if stop.is_set():
                    return
            queues[i].put(None)
        except BaseException as e:   # handed to the consumer
            queues[i].put(e)
    # `threads` files are open at a time, and always the lowest unfinished ones: the reader of file i + threads starts when file i has
    # been consumed.  (A semaphore any reader may take let readers of later files hold every permit while they waited on their full
    # queues, and the reader of the file the consumer was waiting for never got one.)
    workers = [threading.Thread(target=reader, args=(i,), daemon=True) for i in range(len(paths))]
    for w in workers[:threads]:
        w.start()
    try:
        for i in range(len(paths)):
            while True:
                item = queues[i].get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield item
            if i + threads < len(paths):
                workers[i + threads].start()
    finally:
        stop.set()


def read_records(path: str):
    """The whole file at once: (buffer uint8, [(offset, length)])."""
    bufs, recs, base = [], [], 0
    for b, off, ln in iter_chunks(path):
        bufs.append(b)
        recs += [(int(o) + base, int(l)) for o, l in zip(off, ln)]
        base += b.size
    return (np.concatenate(bufs) if bufs else np.zeros(0, dtype=np.uint8)), recs


def read_records_py(path: str):
    """kseq_read over one file (util/kseq.h:178-222) in Python: returns (buffer uint8, [(offset, length)]) -- the sequences are
    copied back to back into the buffer, line breaks removed.  Reads the whole file; the cross-check of the native reader."""
    with open(path, "rb") as f:
        magic = f.read(2)
    opener = gzip.open if magic == b"\x1f\x8b" else open
    with opener(path, "rb") as f:
        data = f.read()
    out = bytearray()
    recs = []
    n = len(data)
    pos = 0

    def line_end(p):
        e = data.find(b"\n", p)
        return n if e < 0 else e

    # find the first header
    while pos < n and data[pos] not in (0x3e, 0x40):  # '>' '@'
        pos += 1
    while pos < n:
        e = line_end(pos)            # header line (name / comment are not used by seq_dump)
        pos = min(e + 1, n)
        start = len(out)
        c = -1
        while pos < n:               # sequence lines until a line that starts with '>', '+' or '@'
            c = data[pos]
            if c in (0x3e, 0x2b, 0x40):
                break
            if c == 0x0a:
                pos += 1
                c = -1
                continue
            e = line_end(pos)
            line = data[pos:e]
            if line.endswith(b"\r"):
                line = line[:-1]
            out += line
            pos = min(e + 1, n)
            c = -1
        seq_len = len(out) - start
        if c != 0x2b:                # FASTA record (or end of file)
            recs.append((start, seq_len))
            if c == -1:
                break
            continue                 # pos sits on the next header
        # FASTQ: skip the '+' line, read quality lines until they cover the sequence
        e = line_end(pos)
        pos = min(e + 1, n)
        qual_len = 0
        got_line = False
        while pos < n or not got_line:
            if pos >= n:
                break
            e = line_end(pos)
            line = data[pos:e]
            if line.endswith(b"\r"):
                line = line[:-1]
            qual_len += len(line)
            pos = min(e + 1, n)
            got_line = True
            if qual_len >= seq_len:
                break
        if qual_len != seq_len:      # kseq_read returns -2: the caller's loop stops, the record is not used
            del out[start:]
            break
        recs.append((start, seq_len))
        while pos < n and data[pos] not in (0x3e, 0x40):   # next header (last_char == 0: scan for it)
            pos += 1
    return np.frombuffer(bytes(out), dtype=np.uint8), recs


class _Out:
    def __init__(self, prefix2, prefix_idx, cnt):
        self.f2 = open("%s%03d.2bit" % (prefix2, cnt), "wb")
        self.fi = open("%s%03d.idx" % (prefix_idx, cnt), "w")
        self.f2.write(bytes([0, 254]))   # init_seq_mode (lib/bseq.c:93-97)
        self.offset = 2
        self.length = 0

    def put(self, rid, n, words):
        self.fi.write("%u\t%u\t%u\n" % (rid, self.offset + 8, n))
        self.f2.write(np.asarray([rid, n], dtype=np.uint32).tobytes())
        self.f2.write(words.tobytes())
        self.offset += 8 + 4 * words.size

    def close(self):
        self.f2.close()
        self.fi.close()


def run(argv) -> int:
    import getopt
    opts, args = getopt.getopt(argv, "f:s:b:n:d:")
    o = dict(opts)
    if len(args) < 1 or "-d" not in o or "-n" not in o or "-s" not in o or "-f" not in o:
        sys.stderr.write("Usage: seq_dump -f min_read_len -s min_seed_len -b block_size -n seed_files -d outdir input.fofn\n")
        return 1
    flt, seed_flt = parse_num(o["-f"]), parse_num(o["-s"])
    if seed_flt <= flt:
        sys.stderr.write("Error! Seed filter length should be larger than filter length!\n")
        return 1
    block = parse_num(o.get("-b", "0")) or (1 << 64) - 1
    seed_n = int(o["-n"])
    d = o["-d"]
    os.makedirs(d, exist_ok=True)
    part_pre, part_idx = os.path.join(d, "input.part."), os.path.join(d, ".input.part.")
    seed_pre, seed_idx = os.path.join(d, "input.seed."), os.path.join(d, ".input.seed.")
    seeds = [_Out(seed_pre, seed_idx, i + 1) for i in range(seed_n)]
    part_cnt = 1
    part = _Out(part_pre, part_idx, part_cnt)
    next_id, seed_cnt = 0, 1
    fofn = args[0]
    base = os.path.dirname(fofn) or "."
    with open(fofn) as f:
        lines = f.read().split("\n")
    paths = [line if line.startswith("/") else os.path.join(base, line) for line in lines if len(line) and not line.startswith("#")]
    for chunk in iter_chunks_files(paths, chunk_bases=CHUNK_BASES, chunk_recs=CHUNK_RECS):
        next_id, seed_cnt, part, part_cnt = _put_chunk(chunk, flt, seed_flt, block, seeds, seed_n, part, part_cnt, part_pre, part_idx,
                                                       next_id, seed_cnt)
    for s in seeds:
        s.close()
    part.close()
    return 0


def _put_chunk(chunk, flt, seed_flt, block, seeds, seed_n, part, part_cnt, part_pre, part_idx, next_id, seed_cnt):
    """The body of split_data()'s loop (util/seq_dump.c:74-114) for the reads of one chunk."""
    buf, offs, lns = chunk
    sel = ((lns >= flt) & (lns < seed_flt)) | ((lns >= seed_flt) & (lns < LEN_LIMIT))
    keep = [(int(s), int(l)) for s, l in zip(offs[sel], lns[sel])]
    if not keep:
        return next_id, seed_cnt, part, part_cnt
    a_off = np.asarray([s for s, _ in keep], dtype=np.uint64)
    lens = np.asarray([min(l, LEN_LIMIT) for _, l in keep], dtype=np.uint32)   # convert_2bit truncates (seq_dump.c:38)
    words, w_off = overlap.pack_2bit(buf, a_off, lens)                         # one device launch per chunk
    for k, (_, l) in enumerate(keep):
        n = int(lens[k])
        w = words[int(w_off[k]): int(w_off[k]) + (n + 15) // 16]
        if l < seed_flt:
            part.length += l
            if part.length > block:
                part.close()
                part_cnt += 1
                part = _Out(part_pre, part_idx, part_cnt)
                part.length = l
            part.put(next_id, n, w)
        else:
            if seed_cnt > seed_n:
                seed_cnt = 1
            seeds[seed_cnt - 1].put(next_id, n, w)
            seeds[seed_cnt - 1].length += l
            seed_cnt += 1
        next_id += 1
    return next_id, seed_cnt, part, part_cnt


if __name__ == "__main__":
    sys.exit(run(sys.argv[1:]))
