"""ctypes view of oracle/mm_oracle.c (the CPU restatement of `minimap2-nd --step 1`) + helpers that
run the compiled reference binary.  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFDIR = os.path.join(ROOT, "oracle", "_ref")


class MMOpt(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("k", "w", "hpc", "no_diag", "no_dual", "min_cnt", "min_sc", "bw", "max_gap",
                                         "max_skip", "max_iter", "minlen", "seed", "dvt", "maxhan1", "maxhan2", "max_occ")]


class MMReg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("rev", "rid", "qs", "qe", "rs", "re", "mlen", "blen", "score", "cnt", "as_")] \
        + [("hash", C.c_uint32)]


MM128 = np.dtype([("x", np.uint64), ("y", np.uint64)])
REG = np.dtype([(n, np.int32) for n in ("rev", "rid", "qs", "qe", "rs", "re", "mlen", "blen", "score", "cnt", "as_")]
               + [("hash", np.uint32)])


def preset(name: str, dual: bool = False, **kw) -> MMOpt:
    """minimap2/options.c:12-62,84-97 (mm_mapopt_init + mm_set_opt) and main.c:190-193 (--step 1)."""
    o = MMOpt(k=15, w=5, hpc=0, no_diag=1, no_dual=0 if dual else 1, min_cnt=3, min_sc=100, bw=500, max_gap=10000,
              max_skip=25, max_iter=5000, minlen=500, seed=11, dvt=0, maxhan1=5000, maxhan2=500)
    if name == "ava-ont":
        o.bw = 2000
    elif name == "ava-pb":
        o.k, o.hpc = 19, 1
    elif name == "ava-hifi":  # options.c:99-111 (mid_occ_frac 1e-4 is the caller's to pass)
        o.k, o.w, o.hpc = 51, 51, 1
    else:
        raise ValueError(name)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def bind(lib):
    P = C.c_void_p
    lib.nd_mm_sketch.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, P]
    lib.nd_mm_sketch.restype = C.c_int64
    lib.nd_mm_rs_sort128.argtypes = [P, C.c_int64]
    lib.nd_mm_index_build.argtypes = [C.c_int32, P, P, P, P, C.c_int, C.c_int, C.c_int]
    lib.nd_mm_index_build.restype = P
    lib.nd_mm_index_free.argtypes = [P]
    lib.nd_mm_index_n.argtypes = [P]
    lib.nd_mm_index_n.restype = C.c_int64
    lib.nd_mm_index_keys.argtypes = [P]
    lib.nd_mm_index_keys.restype = C.c_int64
    lib.nd_mm_index_dump.argtypes = [P, P, P, P]
    lib.nd_mm_index_mid_occ.argtypes = [P, C.c_float]
    lib.nd_mm_index_mid_occ.restype = C.c_int32
    lib.nd_mm_seeds.argtypes = [P, C.POINTER(MMOpt), C.c_char_p, C.c_int, C.c_int, P, C.c_int64, P, C.c_int]
    lib.nd_mm_seeds.restype = C.c_int64
    lib.nd_mm_chain.argtypes = [C.POINTER(MMOpt), C.c_int64, P, P, P, P, P]
    lib.nd_mm_chain.restype = C.c_int
    lib.nd_mm_read_hash.argtypes = [C.c_char_p, C.c_int, C.c_int]
    lib.nd_mm_read_hash.restype = C.c_uint32
    lib.nd_mm_gen_regs.argtypes = [C.c_uint32, C.c_int, C.c_int, P, P, P]
    lib.nd_mm_gen_regs.restype = C.c_int
    lib.nd_mm_map_read.argtypes = [P, C.POINTER(MMOpt), C.c_int, C.c_uint32, P, C.c_int, P, C.c_int]
    lib.nd_mm_map_read.restype = C.c_int
    lib.nd_mm_step1.argtypes = [C.POINTER(MMOpt), C.c_float, C.c_int, C.c_int32, P, P, P, P, C.c_int32, P, P, P, P, P,
                                C.c_int64, P, P]
    lib.nd_mm_step1.restype = C.c_int64
    lib.nd_mm_step1_mode3.argtypes = lib.nd_mm_step1.argtypes
    lib.nd_mm_step1_mode3.restype = C.c_int64
    return lib


def ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def sketch(lib, codes: np.ndarray, w, k, rid=0, hpc=0) -> np.ndarray:
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    out = np.zeros(codes.size + 1, dtype=MM128)
    n = lib.nd_mm_sketch(ptr(codes), codes.size, w, k, rid, hpc, ptr(out))
    assert n >= 0
    return out[:n].copy()


def step1(lib, opt: MMOpt, tset, qset, mid_occ_frac=2e-4, mid_occ=0, batch_size=None, mode3=False):
    """tset/qset = (ids, lens, codes, off).  Returns (.ovl bytes, mid_occ).  batch_size = the -I value: the
    target set is indexed in parts (nextdenovo_amd.minimap2_nd.index_parts restates mm_idx_gen's rule)."""
    tid, tl, tc, to = tset
    qid, ql, qc, qo = qset
    parts = [(0, tid.size)]
    if batch_size is not None:
        from nextdenovo_amd.minimap2_nd import index_parts
        parts = index_parts(tl, batch_size)
    prev = np.zeros(2, dtype=np.uint32)
    blob = b""
    for lo, hi in parts:
        cap = 1 << 20
        while True:
            out = np.zeros(cap, dtype=np.uint8)
            mo = C.c_int32(0)
            pv = prev.copy()
            n = (lib.nd_mm_step1_mode3 if mode3 else lib.nd_mm_step1)(C.byref(opt), np.float32(mid_occ_frac), mid_occ, hi - lo, ptr(tc), ptr(to[lo:hi]), ptr(tl[lo:hi]),
                                ptr(tid[lo:hi]), qid.size, ptr(qc), ptr(qo), ptr(ql), ptr(qid), ptr(out), cap, C.byref(mo), ptr(pv))
            if n >= 0:
                break
            cap = max(cap * 4, -n * 2)
        prev = pv
        mid_occ = mo.value  # the threshold of the first part is kept (options.c:70-71)
        blob += out[:n].tobytes()
    return blob, mid_occ


class AlnOpt(C.Structure):  # nd_aln_opt (oracle/cigar_oracle.c): the scoring side of -c, defaults of mm_mapopt_init (options.c:36-43)
    _fields_ = [(n, C.c_int32) for n in ("a", "b", "q", "e", "q2", "e2", "sc_ambi", "zdrop", "zdrop_inv", "end_bonus", "min_dp_max", "min_ksw_len")] \
        + [("max_sw_mat", C.c_int64)]


def aln_opt(**kw) -> AlnOpt:
    o = AlnOpt(a=2, b=4, q=4, e=2, q2=24, e2=1, sc_ambi=1, zdrop=400, zdrop_inv=200, end_bonus=-1, min_dp_max=80, min_ksw_len=200,
               max_sw_mat=0)   # (never set by mm_mapopt_init: no cap unless --cap-sw-mem)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def step1_cigar(lib, opt: MMOpt, ao: AlnOpt, tset, qset, mid_occ_frac=2e-4, mid_occ=0, batch_size=None):
    """`--step 1 -c` of the oracle (nd_mm_step1_cigar): .ovl bytes; arguments as step1()."""
    tid, tl, tc, to = tset
    qid, ql, qc, qo = qset
    parts = [(0, tid.size)]
    if batch_size is not None:
        from nextdenovo_amd.minimap2_nd import index_parts
        parts = index_parts(tl, batch_size)
    lib.nd_mm_step1_cigar.restype = C.c_int64
    lib.nd_mm_step1_cigar.argtypes = [C.POINTER(MMOpt), C.POINTER(AlnOpt), C.c_float, C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.c_void_p]
    prev = np.zeros(2, dtype=np.uint32)
    blob = b""
    for lo, hi in parts:
        cap = 1 << 22
        while True:
            out = np.zeros(cap, dtype=np.uint8)
            mo = C.c_int32(0)
            pv = prev.copy()
            n = lib.nd_mm_step1_cigar(C.byref(opt), C.byref(ao), np.float32(mid_occ_frac), mid_occ, hi - lo, ptr(tc), ptr(to[lo:hi]), ptr(tl[lo:hi]),
                                      ptr(tid[lo:hi]), qid.size, ptr(qc), ptr(qo), ptr(ql), ptr(qid), ptr(out), cap, C.byref(mo), ptr(pv))
            if n >= 0:
                break
            cap = max(cap * 4, -n * 2)
        prev = pv
        mid_occ = mo.value
        blob += out[:n].tobytes()
    return blob, mid_occ


def step2_mode0(lib, opt: MMOpt, tset, qsets, minide=0.05, minmatch=100, mid_occ_frac=2e-4):
    return step2(lib, opt, tset, qsets, 0, minide, minmatch, mid_occ_frac)


def step2(lib, opt: MMOpt, tset, qsets, mode=2, minide=0.05, minmatch=100, mid_occ_frac=2e-4, kn=17, wn=10, cn=20, mid_occ_fixed=0):
    """`minimap2-nd --step 2 [--mode 0|2] target query...` with the oracle (mode 2 = the default, minimap2/options.c:56): returns
    (.ovl bytes incl. the 00 FF header, .bl text)."""
    lib.nd_mm_step2.argtypes = [C.POINTER(MMOpt), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int32, C.c_float, C.c_int, C.c_int32,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.nd_mm_step2.restype = C.c_int64
    lib.nd_s2_new.restype = C.c_void_p
    lib.nd_s2_free.argtypes = [C.c_void_p]
    lib.nd_s2_out_bl.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    lib.nd_s2_out_bl.restype = C.c_int64
    tid, tl, tc, to = tset
    st = lib.nd_s2_new()
    prev = np.zeros(2, dtype=np.uint32)
    blob = bytes([0, 255])
    try:
        for qid, ql, qc, qo in qsets:
            cap = 1 << 22
            while True:
                out = np.zeros(cap, dtype=np.uint8)
                mo = C.c_int32(0)
                n = lib.nd_mm_step2(C.byref(opt), mode, kn, wn, cn, np.float32(minide), minmatch, np.float32(mid_occ_frac), int(mid_occ_fixed), tid.size, ptr(tc), ptr(to),
                                          ptr(tl), ptr(tid), qid.size, ptr(qc), ptr(qo), ptr(ql), ptr(qid), ptr(out), cap, C.byref(mo),
                                          ptr(prev), st)
                if n >= 0:
                    break
                raise RuntimeError("output buffer too small (the filter state has already advanced)")
            blob += out[:n].tobytes()
        buf = C.create_string_buffer(1 << 24)
        nb = lib.nd_s2_out_bl(st, buf, len(buf))
        return blob, buf.raw[:nb].decode()
    finally:
        lib.nd_s2_free(st)


def load_set(path):
    """One .2bit file -> (ids, lens, codes, off) as the oracle wants them."""
    from nextdenovo_amd import ovl
    ids, lens, words, woff = ovl.read_2bit(path)
    codes, off = ovl.unpack_codes(words, woff, lens)
    return (np.ascontiguousarray(ids), np.ascontiguousarray(lens), codes, off)


def ref_step1(target_2bit, query_2bit, out_path, preset_name="ava-ont", dual=False, extra=(), threads=3):
    cmd = [os.path.join(REFDIR, "minimap2-nd"), "--step", "1"]
    if dual:
        cmd.append("--dual=yes")
    cmd += ["-t", str(threads), "-x", preset_name, *extra, target_2bit, query_2bit, "-o", out_path]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with open(out_path, "rb") as f:
        return f.read()


def dump_reads(workdir, seqs_ascii, seed_cutoff, read_cutoff=500):
    """seq_dump on a FASTA -> (seed.2bit, part.2bit or None)."""
    from refpipe import run, write_fasta
    os.makedirs(workdir, exist_ok=True)
    fa = os.path.join(workdir, "reads.fa")
    write_fasta(fa, seqs_ascii)
    fofn = os.path.join(workdir, "input.fofn")
    with open(fofn, "w") as f:
        f.write(fa + "\n")
    db = os.path.join(workdir, "db")
    os.makedirs(db, exist_ok=True)
    run([os.path.join(REFDIR, "seq_dump"), "-f", str(read_cutoff), "-s", str(seed_cutoff), "-b", "2g", "-n", "1", "-d", db,
         fofn])
    seed = os.path.join(db, "input.seed.001.2bit")
    part = os.path.join(db, "input.part.001.2bit")
    if not (os.path.exists(part) and os.path.getsize(part) > 2):
        part = None
    return seed, part
