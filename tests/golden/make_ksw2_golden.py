#!/usr/bin/env python
"""Golden vectors for the two-piece affine-gap extension kernel: fuzzed problems (tests/ksw_util.py:problems) and what the compiled
reference function `ksw_extd2_sse` (oracle/_ref/libksw2ref.so, built from minimap2/ksw2_extd2_sse.c) returns for them -- every
field of ksw_extz_t and the CIGAR.  Run in the build container (needs oracle/_ref):  python tests/golden/make_ksw2_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ksw_util as K  # noqa: E402

SEED, COUNT, MAX_LEN = 20260925, 600, 260


def write(ref, ps, name):
    res, cig, cig_off = [], [], [0]
    for p in ps:
        r, c = K.call_sse(ref, p["q"], p["t"], p["mat"], *p["gaps"], p["w"], p["zdrop"], p["end_bonus"], p["flag"])
        res.append(r)
        cig.extend(c)
        cig_off.append(len(cig))
    np.savez_compressed(os.path.join(HERE, name), res=np.asarray(res, dtype=np.int64), cigar=np.asarray(cig, dtype=np.uint32),
                        cigar_off=np.asarray(cig_off, dtype=np.int64))
    print(name, len(ps), "problems,", len(cig), "cigar operations,", os.path.getsize(os.path.join(HERE, name)), "bytes")


def main():
    ref = C.CDLL(K.REF)
    write(ref, K.problems(SEED, COUNT, MAX_LEN), "ksw2.npz")
    write(ref, K.mid_problems(), "ksw2_mid.npz")   # targets of 1,025 .. 4,096 bases (the kernel's large LDS tier)
    # ksw_ll_i16 (oracle/_ref/libksw2llref.so, built from minimap2/ksw2_ll_sse.c): score, query end, target end
    ll = C.CDLL(K.REF_LL)
    res = np.asarray([K.call_ll_ref(ll, p) for p in K.ll_problems()], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "ksw_ll.npz"), res=res)
    print("ksw_ll.npz", res.shape[0], "problems")


if __name__ == "__main__":
    main()
