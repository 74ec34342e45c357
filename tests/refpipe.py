"""Test infrastructure: drive the REAL reference binaries built under oracle/_ref
(see oracle/Makefile) through the correction-stage chain by hand, exactly the
commands nextDenovo would write into its *.sh files:

    seq_dump -> minimap2-nd --step 1 (seed x part, seed x seed) -> ovl_sort
    -> pile assembly (lib/nextcorrect.py:92-143, through ovlseq.so) -> nextCorrect

Nothing here is imported by the product (nextdenovo_amd/); only tests/ and
tests/golden/make_golden.py use it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFDIR = os.path.join(os.path.dirname(HERE), "oracle", "_ref")


def have_ref(*names) -> bool:
    names = names or ("nextcorrect.so",)
    return all(os.path.exists(os.path.join(REFDIR, n)) for n in names)


class ConsensusTrimed(C.Structure):
    _fields_ = [("len", C.c_uint), ("identity", C.c_float), ("seq", C.c_void_p)]


_CNS = None


def ref_cns():
    global _CNS
    if _CNS is None:
        lib = C.CDLL(os.path.join(REFDIR, "nextcorrect.so"))
        lib.nextCorrect.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_uint,
                                    C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_float, C.c_uint,
                                    C.c_uint, C.c_int]
        lib.nextCorrect.restype = C.POINTER(ConsensusTrimed)
        lib.free_consensus_trimed.argtypes = [C.POINTER(ConsensusTrimed)]
        _CNS = lib
    return _CNS


def call_nextcorrect(lib, seqs, aln_start, aln_end, max_aln_length, min_len_aln=500, max_cov_aln=130,
                     min_cov_base=4, max_lq_length=10000, ratio=0.8, split=0, fast=0, read_type=1):
    """Mirror of lib/nextcorrect.py:72-90 correct().  Returns (len, identity, seq bytes or None)."""
    n = len(seqs)
    c_seqs = (C.c_char_p * n)()
    c_seqs[:] = seqs
    st = (C.c_uint * n)(*aln_start)
    en = (C.c_uint * n)(*aln_end)
    r = lib.nextCorrect(c_seqs, st, en, n, max_aln_length, min_len_aln, max_cov_aln, min_cov_base,
                        max_lq_length, ratio, split, fast, read_type)
    ln = r.contents.len
    ide = r.contents.identity
    # len 2/3 buffers are uninitialised in the reference (nextcorrect.c:261-266): do not read them
    seq = C.string_at(r.contents.seq, ln) if ln > 4 else None
    lib.free_consensus_trimed(r)
    return ln, ide, seq


def write_fasta(path, seqs_ascii):
    with open(path, "wb") as f:
        for i, s in enumerate(seqs_ascii):
            f.write(b">r%d\n" % i)
            f.write(s)
            f.write(b"\n")


def run(cmd, cwd=None):
    subprocess.run(cmd, cwd=cwd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def run_overlap_chain(workdir, fasta, seed_cutoff, read_cutoff=500, preset="ava-ont", threads=4,
                      sort_depth=40, extra=()):
    """Returns (idxs_fofn, sorted_ovl)."""
    db = os.path.join(workdir, "db")
    ra = os.path.join(workdir, "ra")
    os.makedirs(db, exist_ok=True)
    os.makedirs(ra, exist_ok=True)
    fofn = os.path.join(workdir, "input.fofn")
    with open(fofn, "w") as f:
        f.write(fasta + "\n")
    R = lambda n: os.path.join(REFDIR, n)
    run([R("seq_dump"), "-f", str(read_cutoff), "-s", str(seed_cutoff), "-b", "2g", "-n", "1", "-d", db, fofn])
    seed2 = os.path.join(db, "input.seed.001.2bit")
    part2 = os.path.join(db, "input.part.001.2bit")
    ovls = []
    if os.path.exists(part2) and os.path.getsize(part2) > 2:
        o = os.path.join(ra, "input.seed.001.2bit.0.ovl")
        run([R("minimap2-nd"), "--step", "1", "--dual=yes", "-t", str(threads), "-x", preset, *extra, seed2, part2, "-o", o])
        ovls.append(o)
    o = os.path.join(ra, "input.seed.001.2bit.1.ovl")
    run([R("minimap2-nd"), "--step", "1", "-I", "3G", "-t", str(threads), "-x", preset, *extra, seed2, seed2, "-o", o])
    ovls.append(o)
    with open(os.path.join(ra, "input.fofn"), "w") as f:
        f.write("\n".join(ovls) + "\n")
    sorted_ovl = os.path.join(ra, "input.seed.001.sorted.ovl")
    run([R("ovl_sort"), "-m", "2g", "-t", str(max(2, threads)), "-k", str(sort_depth), "-i",
         os.path.join(db, ".input.seed.001.idx"), "-o", os.path.basename(sorted_ovl), "input.fofn"], cwd=ra)
    idxs = os.path.join(workdir, "idxs.fofn")
    with open(idxs, "w") as f:
        for n in sorted(os.listdir(db)):
            if n.startswith(".input.") and n.endswith(".idx"):
                f.write(os.path.join(db, n) + "\n")
    return idxs, sorted_ovl


class _Ids(C.Structure):
    _fields_ = [("prev_qname", C.c_uint32), ("prev_tname", C.c_uint32)]


class _Ovl(C.Structure):
    _fields_ = [("ovl", C.c_void_p), ("f2bits", C.c_void_p), ("f2bits_", C.c_void_p),
                ("decode_tbl", C.POINTER(C.c_uint32)), ("handle_index", C.c_int), ("indexs", C.c_void_p),
                ("ovlbuf", C.c_void_p), ("bitbuf", C.c_void_p)]


def read_piles(idxs, sorted_ovl, min_len_seed=0, min_len_aln=500, max_cov_aln=130, min_cov_seed=10,
               blacklist=None):
    """Pile assembly exactly as lib/nextcorrect.py:92-143 + worker :183-199, through the
    reference ovlseq.so.  Yields (seed, seqs[bytes], aln_start, aln_end, max_aln_length, recs)."""
    OVL = C.CDLL(os.path.join(REFDIR, "ovlseq.so"))
    OVL.init_ovls.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    OVL.init_ovls.restype = C.POINTER(_Ovl)
    OVL.destory_ovls.argtypes = [C.POINTER(_Ovl)]
    OVL.decode_ovl.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(_Ids), C.POINTER(C.c_uint32),
                               C.c_void_p, C.c_int]
    OVL.decode_ovl.restype = C.c_int
    OVL.getseq.argtypes = [C.POINTER(_Ovl), C.POINTER(C.c_uint32)]
    OVL.getseq.restype = C.c_void_p
    db = OVL.init_ovls(idxs.encode(), sorted_ovl.encode(), 1)
    blacklist = blacklist or set()
    ids_ = _Ids(0, 0)
    arr = (C.c_uint32 * 8)()

    def emit(seed_name, recs, aln_start, aln_end, max_aln_length):
        seqs = []
        for r in recs:
            a = (C.c_uint32 * 8)(*r)
            seqs.append(C.string_at(OVL.getseq(db, a)))
        return (seed_name, seqs, aln_start, aln_end, max_aln_length, np.asarray(recs, dtype=np.uint32))

    count = total_length = seed_length = max_aln_length = 0
    used, seed_name, recs, aln_start, aln_end = set(), '', [], [], []
    last_seed = -1
    while OVL.decode_ovl(db.contents.ovl, db.contents.decode_tbl, C.byref(ids_), arr, db.contents.ovlbuf, 8) >= 0:
        t_name, _, t_s, t_e, q_name, q_s, q_e, match = list(arr)
        if seed_name == '+' or (last_seed != -1 and t_name != last_seed):
            if seed_length and total_length / seed_length >= min_cov_seed and seed_name != '+':
                yield emit(seed_name, recs, aln_start, aln_end, max_aln_length)
            used, seed_name, recs, aln_start, aln_end = set(), '', [], [], []
            total_length = seed_length = max_aln_length = 0
        if seed_name == '':
            seed_length = t_e + 1
            total_length = 0
            max_aln_length = seed_length
            seed_name = t_name if seed_length >= min_len_seed and t_name not in blacklist else '+'
        if t_e - t_s < min_len_aln or total_length / seed_length > max_cov_aln * 1.5 or q_name in used \
                or seed_name == '+':
            continue
        recs.append(list(arr))
        used.add(q_name)
        aln_start.append(t_s)
        aln_end.append(t_e)
        total_length += t_e - t_s + 1
        v = t_e - t_s + q_e - q_s + 2
        if v > max_aln_length and t_name != q_name:
            max_aln_length = v
        last_seed = t_name
    if seed_length and total_length / seed_length >= min_cov_seed and seed_name != '+':
        yield emit(seed_name, recs, aln_start, aln_end, max_aln_length)
    OVL.destory_ovls(db)
