"""ctypes view of oracle/ovlsort_oracle.c + a runner of the compiled reference ovl_sort.  Test infrastructure."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFDIR = os.path.join(ROOT, "oracle", "_ref")
OVL = np.dtype([(n, np.uint32) for n in ("rev", "qname", "qs", "qe", "tname", "ts", "te", "match")])


def bind(lib):
    P = C.c_void_p
    lib.nd_os_expand.argtypes = [P, C.c_int64, P, C.c_uint32, P]
    lib.nd_os_expand.restype = C.c_int64
    lib.nd_os_order.argtypes = [P, C.c_int64, P]
    lib.nd_os_filter.argtypes = [P, P, C.c_int64, P, C.c_int, C.c_int, C.c_int, P, P, P, P]
    lib.nd_os_filter.restype = C.c_int64
    lib.nd_os_filter2.argtypes = [P, P, C.c_int64, P, C.c_int, C.c_int, C.c_int, P, P, P, P, C.c_int]
    lib.nd_os_filter2.restype = C.c_int64
    lib.nd_os_encode.argtypes = [P, C.c_int64, P]
    lib.nd_os_encode.restype = C.c_int64
    return lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def read_idx(path):
    """`.idx` of a seed file -> (seed_len table indexed by id, min seed length)."""
    tab = np.loadtxt(path, dtype=np.int64, ndmin=2)
    n = int(tab[:, 0].max()) + 1 if tab.size else 0
    sl = np.zeros(n, dtype=np.uint32)
    sl[tab[:, 0]] = tab[:, 2].astype(np.uint32)
    return sl, int(tab[:, 2].min()) if tab.size else 0


def oracle_sort(lib, raw_files, seed_len, min_seed_len, max_bin_cov=40, flank=300, hq=False):
    """raw_files: list of uint32[n,8] arrays (decode_ovl order), one per input .ovl in fofn order.
    Returns (sorted.ovl bytes, .bl text)."""
    cands = []
    for raw in raw_files:
        raw = np.ascontiguousarray(raw, dtype=np.uint32)
        out = np.zeros(2 * max(1, raw.shape[0]), dtype=OVL)
        m = lib.nd_os_expand(ptr(raw), raw.shape[0], ptr(seed_len), seed_len.size, ptr(out))
        cands.append(out[:m])
    cand = np.concatenate(cands) if cands else np.zeros(0, dtype=OVL)
    n = cand.size
    perm = np.zeros(max(1, n), dtype=np.uint32)
    lib.nd_os_order(ptr(cand), n, ptr(perm))
    n_seeds = int((seed_len > 0).sum())
    out = np.zeros(n + n_seeds + 1, dtype=OVL)
    bl_id = np.zeros(n_seeds + 1, dtype=np.uint32)
    bl_kind = np.zeros(n_seeds + 1, dtype=np.uint8)
    n_bl = C.c_int64(0)
    n_out = lib.nd_os_filter2(ptr(cand), ptr(perm), n, ptr(seed_len), max_bin_cov, flank, min_seed_len, ptr(out), ptr(bl_id),
                              ptr(bl_kind), C.byref(n_bl), 1 if hq else 0)
    buf = np.zeros(40 * max(1, n_out), dtype=np.uint8)
    nb = lib.nd_os_encode(ptr(out), n_out, ptr(buf))
    bl = "".join("%d %s\n" % (int(bl_id[i]), chr(int(bl_kind[i]))) for i in range(n_bl.value))
    return buf[:nb].tobytes(), bl, out[:n_out].copy()


def ref_sort(workdir, idx, ovl_files, k=40, threads=2, mem="2g", flank=None, hq=False):
    """Run oracle/_ref/ovl_sort; returns (sorted.ovl bytes, .bl text)."""
    fofn = os.path.join(workdir, "sort.fofn")
    with open(fofn, "w") as f:
        f.write("\n".join(ovl_files) + "\n")
    out = "sorted.%d.ovl" % os.getpid()
    cmd = [os.path.join(REFDIR, "ovl_sort"), "-m", mem, "-t", str(threads), "-k", str(k), "-i", idx, "-o", out]
    if flank is not None:
        cmd += ["-l", str(flank)]
    if hq:
        cmd.append("-H")
    cmd.append(fofn)
    subprocess.run(cmd, cwd=workdir, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with open(os.path.join(workdir, out), "rb") as f:
        blob = f.read()
    blp = os.path.join(workdir, out + ".bl")
    bl = open(blp).read() if os.path.exists(blp) else ""
    return blob, bl
