"""The ksw2 oracle (oracle/ksw2_oracle.c: scalar restatement of minimap2's `ksw_extd2_sse`) against the compiled reference function
and against the committed vectors the reference produced (tests/golden/ksw2.npz: the problems are regenerated from the seed)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import ksw_util as K  # noqa: E402
from make_ksw2_golden import COUNT, MAX_LEN, SEED  # noqa: E402


def golden():
    d = np.load(os.path.join(HERE, "golden", "ksw2.npz"))
    ps = K.problems(SEED, COUNT, MAX_LEN)
    want = []
    for i in range(len(ps)):
        want.append((tuple(int(x) for x in d["res"][i]), tuple(int(x) for x in d["cigar"][d["cigar_off"][i]:d["cigar_off"][i + 1]])))
    return ps, want


def golden_mid():
    d = np.load(os.path.join(HERE, "golden", "ksw2_mid.npz"))
    ps = K.mid_problems()
    want = [(tuple(int(x) for x in d["res"][i]), tuple(int(x) for x in d["cigar"][d["cigar_off"][i]:d["cigar_off"][i + 1]]))
            for i in range(len(ps))]
    return ps, want


def test_oracle_matches_mid_length_golden_vectors(oracle_lib):
    """Targets of 1,025 .. 4,096 bases (the device kernel's large LDS tier): the oracle against the compiled reference's vectors."""
    ps, want = golden_mid()
    assert min(p["t"].size for p in ps) > 1024 and max(p["t"].size for p in ps) == 4096
    for i, (p, w) in enumerate(zip(ps, want)):
        got = K.call_oracle(oracle_lib, p["q"], p["t"], p["mat"], *p["gaps"], p["w"], p["zdrop"], p["end_bonus"], p["flag"])
        assert got == w, i
    assert sum(1 for r, _ in want if r[1]) >= 5             # z-dropped
    assert sum(1 for r, c in want if len(c) > 20) >= 30     # long CIGARs


def test_oracle_matches_golden_vectors(oracle_lib):
    ps, want = golden()
    flags = 0
    for i, (p, w) in enumerate(zip(ps, want)):
        got = K.call_oracle(oracle_lib, p["q"], p["t"], p["mat"], *p["gaps"], p["w"], p["zdrop"], p["end_bonus"], p["flag"])
        assert got == w, i
        flags |= p["flag"]
    assert flags == 0xdf                                   # every flag of the kernel occurs
    assert sum(1 for r, _ in want if r[1]) > 30            # z-dropped problems
    assert sum(1 for r, c in want if len(c) > 3) > 60      # CIGARs with indels
    assert sum(1 for r, _ in want if r[10]) > 20           # reach_end


@pytest.mark.skipif(not os.path.exists(K.REF), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed,count,max_len", [(3, 2500, 300), (4, 250, 2500)])
def test_oracle_matches_live_reference(oracle_lib, seed, count, max_len):
    ref = C.CDLL(K.REF)
    for i, p in enumerate(K.problems(seed, count, max_len)):
        a = K.call_sse(ref, p["q"], p["t"], p["mat"], *p["gaps"], p["w"], p["zdrop"], p["end_bonus"], p["flag"])
        b = K.call_oracle(oracle_lib, p["q"], p["t"], p["mat"], *p["gaps"], p["w"], p["zdrop"], p["end_bonus"], p["flag"])
        assert a == b, i


def test_ll_oracle_matches_golden_vectors(oracle_lib):
    """nd_oracle_ksw_ll_i16 (the striped local-alignment score of the -c path's inversion test) against the vectors of the compiled
    ksw_ll_qinit + ksw_ll_i16 (tests/golden/ksw_ll.npz)."""
    want = np.load(os.path.join(HERE, "golden", "ksw_ll.npz"))["res"]
    ps = K.ll_problems()
    assert len(ps) == want.shape[0]
    for i, p in enumerate(ps):
        assert K.call_ll_oracle(oracle_lib, p) == tuple(int(x) for x in want[i]), i
    assert (want[:, 0] > 100).sum() > 30 and (want[:, 0] == 0).sum() >= 1
    # the padding columns behind the query take part: ends beyond the query's last base occur
    assert sum(1 for p, w in zip(ps, want) if w[1] >= p["q"].size) >= 1


@pytest.mark.skipif(not os.path.exists(K.REF_LL), reason="oracle/_ref not built")
def test_ll_oracle_matches_live_reference(oracle_lib):
    ref = C.CDLL(K.REF_LL)
    for i, p in enumerate(K.ll_problems(seed=11, n=1200)):
        assert K.call_ll_oracle(oracle_lib, p) == K.call_ll_ref(ref, p), i


REF_Z = os.path.join(os.path.dirname(K.REF), "libksw2zref.so")


@pytest.mark.skipif(not os.path.exists(REF_Z), reason="oracle/_ref not built")
def test_one_gap_piece_is_the_two_piece_kernel_with_equal_pieces(oracle_lib):
    """`-c -O a -E b` (q == q2, e == e2): the reference aligns with ksw_extz2_sse (minimap2/align.c:313-331), the product with its
    two-piece kernel.  Under the four flag sets the -c path passes (align.c:697,733,745,764,821) the compiled ksw2_extz2_sse.c and the
    two-piece restatement with equal pieces return the same scores, end points and CIGARs on fuzzed problems.  (They differ under
    KSW_EZ_APPROX_DROP -- the first anti-diagonal's z-drop update -- which that path never sets.)"""
    refz = C.CDLL(REF_Z)
    fz = refz.ksw_extz2_sse
    fz.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int8, C.c_void_p, C.c_int8, C.c_int8, C.c_int, C.c_int, C.c_int, C.c_int,
                   C.POINTER(K.Extz)]
    fz.restype = None
    free = C.CDLL(None).free

    def call_z(p, go, ge, flag):
        ez = K.Extz()
        q, t = np.ascontiguousarray(p["q"], dtype=np.uint8), np.ascontiguousarray(p["t"], dtype=np.uint8)
        fz(None, q.size, q.ctypes.data, t.size, t.ctypes.data, 5, p["mat"].ctypes.data, go, ge, p["w"], p["zdrop"], p["end_bonus"], flag, C.byref(ez))
        d = dict(max=ez.max_zd & 0x7fffffff, zdropped=ez.max_zd >> 31, max_q=ez.max_q, max_t=ez.max_t, mqe=ez.mqe, mqe_t=ez.mqe_t, mte=ez.mte,
                 mte_q=ez.mte_q, score=ez.score, n_cigar=ez.n_cigar, reach_end=ez.reach_end)
        cig = [ez.cigar[i] for i in range(ez.n_cigar)] if ez.n_cigar else []
        if ez.cigar:
            free(ez.cigar)
        return K._as_tuple(d, cig)
    flags = [K.F_EXTZ_ONLY | K.F_RIGHT | K.F_REV_CIGAR, K.F_APPROX_MAX, 0, K.F_EXTZ_ONLY]
    n = 0
    for i, p in enumerate(K.problems(41, 3000, max_len=500)):
        go, ge = p["gaps"][0], p["gaps"][1]
        fl = flags[i % 4]
        assert call_z(p, go, ge, fl) == K.call_oracle(oracle_lib, p["q"], p["t"], p["mat"], go, ge, go, ge, p["w"], p["zdrop"], p["end_bonus"], fl), i
        n += 1
    assert n == 3000
