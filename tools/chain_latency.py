#!/usr/bin/env python
"""Latency of the dependent chains: ONE long alignment on an otherwise idle MI355X through the C ABI's align() -- K7's HIP-event time
and the rest of the call (K8a + transfers) per edit step.  python tools/chain_latency.py [out.json]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402
from nextdenovo_amd import api, synth  # noqa: E402

lib = api.load()
ASC = np.frombuffer(b"ACGT", dtype=np.uint8)
rng = np.random.default_rng(1)
res = []
util.gpu_align(lib, b"ACGTACGTAC" * 50, b"ACGTACGTAC" * 50)   # HIP initialisation
for L in (20000, 100000, 400000, 1000000):
    base = rng.integers(0, 4, L, dtype=np.uint8)
    q = ASC[synth.mutate(base, np.random.default_rng(2), "ont")[0]].tobytes()
    t = ASC[synth.mutate(base, np.random.default_rng(3), "ont")[0]].tobytes()
    best = None
    for rep in range(2):
        api.reset_stats()
        t0 = time.perf_counter()
        n, tu, qu, ts, qs = util.gpu_align(lib, q, t)
        dt = time.perf_counter() - t0
        st = api.stats()
        if best is None or dt < best[0]:
            best = (dt, st["forward_ms"], st["d_steps"], st["cells"], n)
    dt, fwd, d, cells, n = best
    r = {"bases": L, "aln_len": int(n), "d_steps": int(d), "cells_per_step": cells / max(1, d), "call_ms": dt * 1e3, "k7_ms": fwd,
         "k7_us_per_step": fwd * 1e3 / max(1, d), "rest_of_call_us_per_step": (dt * 1e3 - fwd) * 1e3 / max(1, d)}
    res.append(r)
    print(json.dumps(r), flush=True)
if len(sys.argv) > 1:
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
    json.dump(res, open(sys.argv[1], "w"), indent=1)
