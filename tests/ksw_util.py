"""ctypes views of the ksw2 entry points: the compiled reference (oracle/_ref/libksw2ref.so), the oracle, the product."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libksw2ref.so")

F_SCORE_ONLY, F_RIGHT, F_GENERIC_SC, F_APPROX_MAX, F_APPROX_DROP, F_EXTZ_ONLY, F_REV_CIGAR = 0x01, 0x02, 0x04, 0x08, 0x10, 0x40, 0x80


class Extz(C.Structure):  # ksw_extz_t (minimap2/ksw2.h:23-32)
    _fields_ = [("max_zd", C.c_uint32), ("max_q", C.c_int), ("max_t", C.c_int), ("mqe", C.c_int), ("mqe_t", C.c_int), ("mte", C.c_int),
                ("mte_q", C.c_int), ("score", C.c_int), ("m_cigar", C.c_int), ("n_cigar", C.c_int), ("reach_end", C.c_int),
                ("cigar", C.POINTER(C.c_uint32))]


class Result(C.Structure):  # nd_ksw_result (oracle/ksw2_oracle.c)
    _fields_ = [(n, C.c_int32) for n in ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score", "n_cigar", "reach_end")]


FIELDS = ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score", "n_cigar", "reach_end")


def matrix(a=1, b=4, sc_ambi=1):
    """ksw_gen_simple_mat (minimap2/align.c:12-24) for m = 5."""
    m = np.zeros((5, 5), dtype=np.int8)
    m[:4, :4] = -abs(b)
    for i in range(4):
        m[i, i] = a
    m[4, :] = -abs(sc_ambi)
    m[:, 4] = -abs(sc_ambi)
    return np.ascontiguousarray(m.reshape(-1))


def _as_tuple(d, cigar):
    return tuple(int(d[k]) for k in FIELDS), tuple(int(c) for c in cigar)


def call_sse(lib, q, t, mat, gapo, gape, gapo2, gape2, w, zdrop, end_bonus, flag, free=None):
    """`ksw_extd2_sse` with the reference's signature (the compiled reference, or the product's export of the same name)."""
    f = lib.ksw_extd2_sse
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int8, C.c_void_p, C.c_int8, C.c_int8, C.c_int8, C.c_int8, C.c_int,
                  C.c_int, C.c_int, C.c_int, C.POINTER(Extz)]
    f.restype = None
    ez = Extz()
    q, t = np.ascontiguousarray(q, dtype=np.uint8), np.ascontiguousarray(t, dtype=np.uint8)
    f(None, q.size, q.ctypes.data, t.size, t.ctypes.data, 5, mat.ctypes.data, gapo, gape, gapo2, gape2, w, zdrop, end_bonus, flag, C.byref(ez))
    d = dict(max=ez.max_zd & 0x7fffffff, zdropped=ez.max_zd >> 31, max_q=ez.max_q, max_t=ez.max_t, mqe=ez.mqe, mqe_t=ez.mqe_t, mte=ez.mte,
             mte_q=ez.mte_q, score=ez.score, n_cigar=ez.n_cigar, reach_end=ez.reach_end)
    cig = [ez.cigar[i] for i in range(ez.n_cigar)] if ez.n_cigar else []
    if ez.cigar:
        (free or C.CDLL(None).free)(ez.cigar)
    return _as_tuple(d, cig)


def call_oracle(lib, q, t, mat, gapo, gape, gapo2, gape2, w, zdrop, end_bonus, flag):
    f = lib.nd_oracle_ksw_extd2
    f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int8, C.c_void_p, C.c_int8, C.c_int8, C.c_int8, C.c_int8, C.c_int, C.c_int,
                  C.c_int, C.c_int, C.POINTER(Result), C.c_void_p, C.c_int]
    f.restype = C.c_int
    r = Result()
    q, t = np.ascontiguousarray(q, dtype=np.uint8), np.ascontiguousarray(t, dtype=np.uint8)
    cap = q.size + t.size + 4
    cig = np.zeros(cap, dtype=np.uint32)
    f(q.size, q.ctypes.data, t.size, t.ctypes.data, 5, mat.ctypes.data, gapo, gape, gapo2, gape2, w, zdrop, end_bonus, flag, C.byref(r),
      cig.ctypes.data, cap)
    d = {k: getattr(r, k) for k in FIELDS}
    return _as_tuple(d, cig[:r.n_cigar])


def problems(seed, n, max_len=400):
    """Fuzzed extension problems in the shapes mm_align_pair hands over (gap filling between anchors, end extension with z-drop):
    related sequences with substitutions / short and long indels, N bases, all band / z-drop / flag combinations, lengths around
    the multiples of 16."""
    rng = np.random.default_rng(seed)
    out = []
    for it in range(n):
        L = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 48, 64, 100, int(rng.integers(1, max_len))]))
        base = rng.integers(0, 4, L).astype(np.uint8)
        def mutate(x):
            y = []
            i = 0
            rate = float(rng.choice([0.0, 0.02, 0.1, 0.25], p=[.15, .35, .35, .15]))
            while i < x.size:
                r = rng.random()
                if r < rate / 3:
                    y.append(int(rng.integers(0, 4)))
                    i += 1
                elif r < 2 * rate / 3:
                    i += int(rng.choice([1, 1, 2, 5, 30]))
                elif r < rate:
                    y.extend(rng.integers(0, 4, int(rng.choice([1, 1, 2, 5, 30]))).tolist())
                else:
                    y.append(int(x[i]))
                    i += 1
            y = np.asarray(y if y else [0], dtype=np.uint8)
            if rng.random() < 0.15:
                y[rng.integers(0, y.size, max(1, y.size // 20))] = 4
            return y
        q, t = mutate(base), mutate(base)
        if rng.random() < 0.2:
            t = np.concatenate([t, rng.integers(0, 4, int(rng.integers(1, 80))).astype(np.uint8)])
        if rng.random() < 0.1:
            q = rng.integers(0, 4, int(rng.integers(1, 60))).astype(np.uint8)
        w = int(rng.choice([-1, 751, 50, 10, 3, 1, 0], p=[.3, .25, .2, .1, .05, .05, .05]))
        zdrop = int(rng.choice([-1, 400, 100, 10, 0], p=[.35, .3, .2, .1, .05]))
        flag = 0
        for f, pr in ((F_SCORE_ONLY, .2), (F_RIGHT, .4), (F_GENERIC_SC, .2), (F_APPROX_MAX, .25), (F_APPROX_DROP, .3), (F_EXTZ_ONLY, .5),
                      (F_REV_CIGAR, .4)):
            if rng.random() < pr:
                flag |= f
        gaps = [(4, 2, 24, 1), (6, 2, 26, 1), (5, 4, 56, 1), (4, 2, 4, 2), (24, 1, 4, 2), (2, 1, 10, 1)][int(rng.integers(0, 6))]
        mat = matrix(*[(2, 4, 1), (1, 4, 1), (1, 19, 0), (2, 8, 2)][int(rng.integers(0, 4))])
        out.append(dict(q=q, t=t, mat=mat, gaps=gaps, w=w, zdrop=zdrop, end_bonus=int(rng.choice([-1, 0, 5, 50])), flag=flag))
    return out


class Job(C.Structure):  # ndgpu_ksw_job (include/ndgpu_overlap.h)
    _fields_ = [("query", C.c_void_p), ("target", C.c_void_p), ("mat", C.c_void_p)] + \
        [(n, C.c_int32) for n in ("qlen", "tlen", "w", "zdrop", "end_bonus", "flag")] + \
        [(n, C.c_int8) for n in ("m", "gapo", "gape", "gapo2", "gape2")]


class BatchResult(C.Structure):  # ndgpu_ksw_result
    _fields_ = [(n, C.c_int32) for n in FIELDS] + [("cigar", C.POINTER(C.c_uint32))]


def call_batch(lib, ps):
    """ndgpu_ksw_extd2_batch over a list of problems -> list of (fields, cigar)."""
    n = len(ps)
    jobs = (Job * n)()
    keep = []
    for j, p in zip(jobs, ps):
        q, t = np.ascontiguousarray(p["q"], dtype=np.uint8), np.ascontiguousarray(p["t"], dtype=np.uint8)
        keep.append((q, t, p["mat"]))
        j.query, j.target, j.mat = q.ctypes.data, t.ctypes.data, p["mat"].ctypes.data
        j.qlen, j.tlen, j.w, j.zdrop, j.end_bonus, j.flag = q.size, t.size, p["w"], p["zdrop"], p["end_bonus"], p["flag"]
        j.m, (j.gapo, j.gape, j.gapo2, j.gape2) = 5, p["gaps"]
    res = (BatchResult * n)()
    lib.ndgpu_ksw_extd2_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.ndgpu_ksw_extd2_batch.restype = C.c_int
    rc = lib.ndgpu_ksw_extd2_batch(jobs, n, res)
    if rc != 0:
        raise RuntimeError("ndgpu_ksw_extd2_batch failed (%d)" % rc)
    out, free = [], C.CDLL(None).free
    for r in res:
        cig = [r.cigar[i] for i in range(r.n_cigar)] if r.n_cigar else []
        if r.cigar:
            free(r.cigar)
        out.append(_as_tuple({k: getattr(r, k) for k in FIELDS}, cig))
    return out


def long_problems():
    """Targets beyond the LDS budget of the device kernel (4096): banded, as mm_align_pair calls it (bw 751 / zdrop 400 for
    ava-ont), one unbanded."""
    rng = np.random.default_rng(77)
    out = []
    for tl, w, zdrop, flag in ((5000, 751, 400, F_EXTZ_ONLY), (9000, 200, -1, 0), (4500, -1, 100, F_RIGHT | F_REV_CIGAR), (4100, 50, 400, F_APPROX_MAX)):
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = t.copy()
        for k in range(tl // 60):                      # sprinkle substitutions and short indels
            i = int(rng.integers(0, q.size - 3))
            r = rng.random()
            if r < .5:
                q[i] = (q[i] + 1) % 4
            elif r < .75:
                q = np.delete(q, slice(i, i + int(rng.integers(1, 4))))
            else:
                q = np.insert(q, i, rng.integers(0, 4, int(rng.integers(1, 4))))
        out.append(dict(q=q.astype(np.uint8), t=t, mat=matrix(2, 4, 1), gaps=(4, 2, 24, 1), w=w, zdrop=zdrop, end_bonus=5, flag=flag))
    return out


def mid_problems():
    """Targets of 1,025 .. 4,096 bases: the device kernel's large LDS tier (11 bytes a target position in 44 KB of dynamic LDS),
    which neither the short fuzzed problems nor long_problems() reach.  Related sequences at the divergence of raw-read overlaps
    (gap filling / end extension as mm_align1 hands them over), every flag, banded and unbanded, z-drop on and off."""
    rng = np.random.default_rng(4099)
    out = []
    lens = [1025, 1030, 1100, 1500, 1600, 2047, 2048, 2500, 3000, 3333, 4000, 4080, 4090, 4095, 4096]
    flags = [0, F_EXTZ_ONLY, F_RIGHT | F_REV_CIGAR, F_APPROX_MAX, F_APPROX_DROP | F_EXTZ_ONLY, F_SCORE_ONLY, F_GENERIC_SC | F_EXTZ_ONLY,
             F_RIGHT | F_EXTZ_ONLY | F_APPROX_MAX | F_REV_CIGAR]
    for n, tl in enumerate(lens * 8):
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = t.copy()
        rate = [0.01, 0.05, 0.12, 0.2][n % 4]
        for k in range(int(tl * rate)):
            i = int(rng.integers(0, q.size - 3))
            r = rng.random()
            if r < .4:
                q[i] = (q[i] + 1 + int(rng.integers(0, 3))) % 4
            elif r < .7:
                q = np.delete(q, slice(i, i + int(rng.choice([1, 1, 2, 3, 12]))))
            else:
                q = np.insert(q, i, rng.integers(0, 4, int(rng.choice([1, 1, 2, 3, 12]))))
        if n % 7 == 3:
            q = q[: q.size * 2 // 3]                      # the query ends first: mte / reach_end cases
        if n % 11 == 5:
            q[rng.integers(0, q.size, q.size // 50)] = 4   # N bases
        w = [-1, 751, 200, 50, 10][n % 5]
        zdrop = [-1, 400, 100][n % 3]
        gaps = [(4, 2, 24, 1), (6, 2, 26, 1), (5, 4, 56, 1)][(n // 3) % 3]
        out.append(dict(q=q.astype(np.uint8), t=t, mat=matrix(*[(2, 4, 1), (1, 4, 1)][n % 2]), gaps=gaps, w=w, zdrop=zdrop,
                        end_bonus=[-1, 0, 5, 50][n % 4], flag=flags[(n // 2) % len(flags)]))
    return out


# ---- ksw_ll_i16: the striped local-alignment score of the -c path's inversion test (minimap2/ksw2_ll_sse.c) ----
REF_LL = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libksw2llref.so")


class LlJob(C.Structure):  # ndgpu_ll_job (include/ndgpu_overlap.h)
    _fields_ = [("query", C.c_void_p), ("target", C.c_void_p), ("mat", C.c_void_p)] + [(n, C.c_int32) for n in ("qlen", "tlen", "gapo", "gape")]


class LlResult(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("score", "qe", "te")]


def ll_problems(seed=20260926, n=400):
    """Problems in the shapes the inversion test / mm_align1_inv pose: a stretch of one read against the reverse complement of a
    stretch of the other (related or not), lengths 1 .. 3000 around the multiples of 8 (the stripe width), N bases, several scorings."""
    rng = np.random.default_rng(seed)
    out = []
    for it in range(n):
        L = int(rng.choice([1, 7, 8, 9, 15, 16, 17, 63, 64, 65, 200, int(rng.integers(1, 700)), int(rng.integers(700, 3000)) if it % 10 == 0 else 100]))
        base = rng.integers(0, 4, L).astype(np.uint8)
        kind = it % 4
        if kind == 0:    # unrelated
            q, t = base, rng.integers(0, 4, int(rng.integers(1, 2 * L + 2))).astype(np.uint8)
        else:            # related: substitutions and indels, a flank of unrelated sequence on either side
            y, i, rate = [], 0, float(rng.choice([0.0, 0.05, 0.15, 0.3]))
            while i < base.size:
                r = rng.random()
                if r < rate / 3:
                    y.append(int(rng.integers(0, 4))), 
                    i += 1
                elif r < 2 * rate / 3:
                    i += int(rng.choice([1, 1, 2, 6]))
                elif r < rate:
                    y.extend(rng.integers(0, 4, int(rng.choice([1, 1, 2, 6]))).tolist())
                else:
                    y.append(int(base[i]))
                    i += 1
            t = np.asarray(y if y else [0], dtype=np.uint8)
            if kind == 2:
                t = np.concatenate([rng.integers(0, 4, int(rng.integers(0, 40))).astype(np.uint8), t, rng.integers(0, 4, int(rng.integers(0, 40))).astype(np.uint8)])
            q = base
            if kind == 3 and q.size > 4:
                q = q.copy()
                q[rng.integers(0, q.size, max(1, q.size // 25))] = 4
        a, b, amb = [(2, 4, 1), (1, 4, 1), (1, 19, 0), (2, 8, 2)][int(rng.integers(0, 4))]
        gapo, gape = [(4, 2), (6, 2), (24, 1), (2, 1)][int(rng.integers(0, 4))]
        out.append(dict(q=np.ascontiguousarray(q), t=np.ascontiguousarray(t), mat=matrix(a, b, amb), gapo=gapo, gape=gape))
    return out


def call_ll_ref(lib, p):
    """ksw_ll_qinit(0, 2, qlen, query, 5, mat) + ksw_ll_i16 of the compiled reference -> (score, qe, te)."""
    lib.ksw_ll_qinit.restype = C.c_void_p
    lib.ksw_ll_qinit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.ksw_ll_i16.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.ksw_ll_i16.restype = C.c_int
    qp = lib.ksw_ll_qinit(None, 2, p["q"].size, p["q"].ctypes.data, 5, p["mat"].ctypes.data)
    qe, te = C.c_int(0), C.c_int(0)
    sc = lib.ksw_ll_i16(qp, p["t"].size, p["t"].ctypes.data, p["gapo"], p["gape"], C.byref(qe), C.byref(te))
    C.CDLL(None).free(C.c_void_p(qp))
    return int(sc), int(qe.value), int(te.value)


def call_ll_oracle(lib, p):
    f = lib.nd_oracle_ksw_ll_i16
    f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    f.restype = C.c_int
    qe, te = C.c_int(0), C.c_int(0)
    sc = f(p["q"].size, p["q"].ctypes.data, p["t"].size, p["t"].ctypes.data, p["mat"].ctypes.data, p["gapo"], p["gape"], C.byref(qe), C.byref(te))
    return int(sc), int(qe.value), int(te.value)


def call_ll_batch(lib, ps):
    """ndgpu_ksw_ll_batch over a list of problems -> list of (score, qe, te)."""
    n = len(ps)
    jobs, res = (LlJob * n)(), (LlResult * n)()
    for j, p in zip(jobs, ps):
        j.query, j.target, j.mat = p["q"].ctypes.data, p["t"].ctypes.data, p["mat"].ctypes.data
        j.qlen, j.tlen, j.gapo, j.gape = p["q"].size, p["t"].size, p["gapo"], p["gape"]
    lib.ndgpu_ksw_ll_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.ndgpu_ksw_ll_batch.restype = C.c_int
    rc = lib.ndgpu_ksw_ll_batch(jobs, n, res)
    assert rc == 0, rc
    return [(int(r.score), int(r.qe), int(r.te)) for r in res]
