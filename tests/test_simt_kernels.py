"""Kernel LOGIC on a machine without a GPU: the consensus library's own HIP sources, compiled with g++ against the lane-accurate
interpreter under tests/simt (every lane a fibre, wave-wide operations and barriers as rendezvous points), against the reference's
golden vectors and the oracle.  This is test infrastructure: it says nothing about speed, it is not a backend of the product
(nothing under nextdenovo_amd/ can load it), and the `-m gpu` tests remain the parity tests proper -- what it adds is that a
change to a kernel is checked for logic, barrier placement and inter-wave races before it ever reaches a GPU.

The scoring kernel's forced paths read their switches once per process, so they run in child processes."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import util

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "simt"))


@pytest.fixture(scope="module")
def simt_lib():
    import build_simt
    os.environ.setdefault("NDGPU_CONTEXTS", "1")  # one device context: the interpreter brings its own host threads
    return C.CDLL(build_simt.build())


@pytest.fixture()
def simt_api(simt_lib, monkeypatch):
    """nextdenovo_amd.api bound to the interpreted library for the duration of one test."""
    from nextdenovo_amd import api
    monkeypatch.setattr(api, "_LIB", api._bind(simt_lib))
    return api


def test_interpreter_semantics(simt_lib):
    """The interpreter itself: data-parallel-primitive moves, shuffles and ballots against their definitions."""
    simt_lib.simt_selftest.restype = C.c_int
    assert simt_lib.simt_selftest() == 0


def test_align_golden_vectors(simt_lib):
    """K7 / K7-wide / K8a as compiled from ond_kernels.hip: the reference's align()/align_hq() outputs, bit exact, incl. failures,
    the > 250-gap marker and a live band beyond the LDS fast path (same assertions as the GPU test of the same name)."""
    for i, p in enumerate(util.load_pairs()):
        n, tu, qu, ts, qs = util.gpu_align(simt_lib, p["q"], p["t"], p["hq"])
        assert n == p["aln_len"], i
        if n > 2:
            assert np.array_equal(util.strings_to_ops(ts, qs), p["ops"]), i
            assert (tu, qu) == (p["t_used"], p["q_used"]), i


def test_align_fuzz_vs_oracle(simt_lib, oracle_lib):
    from nextdenovo_amd import synth
    rng = np.random.default_rng(77)
    ok = 0
    for it in range(40):
        L = int(rng.integers(1, 2500))
        base = rng.integers(0, 4 if it % 9 else 2, L, dtype=np.uint8)
        prof = ("ont", "clr", "hifi")[it % 3]
        q = synth.mutate(base, np.random.default_rng(3 * it), prof)[0]
        t = synth.mutate(base, np.random.default_rng(3 * it + 1), prof)[0]
        if it % 10 == 0:
            q = q[int(rng.integers(0, 40)):]
        hq = int(it % 4 == 0)
        qa, ta = util.ASC[q].tobytes(), util.ASC[t].tobytes()
        o, ots, oqs, _ = util.oracle_align(oracle_lib, qa, ta, hq)
        n, tu, qu, ts, qs = util.gpu_align(simt_lib, qa, ta, hq)
        assert n == o.aln_len, it
        if o.status == 1:
            assert ts == ots and qs == oqs and (tu, qu) == (o.t_used, o.q_used), it
            ok += 1
    assert ok > 20


def test_align_beyond_the_sequence_window(simt_lib, oracle_lib):
    """K7 reads its sequences through a 2,048-base ring in LDS that follows the front (chunks of 1,024 bases, the next one in
    flight), and falls back to global memory for a cell outside it: alignments several windows long, with exact runs longer than a
    compare round (64) and than a chunk, a query that starts far into the target's word grid, very unequal error rates (fronts that
    lean to one side) and lengths around the chunk boundaries -- all against the oracle."""
    from nextdenovo_amd import synth
    rng = np.random.default_rng(5)
    cases = []
    for L in (2040, 2049, 3071, 3073, 4100, 9000):
        base = rng.integers(0, 4, L, dtype=np.uint8)
        cases.append((synth.mutate(base, np.random.default_rng(L), "ont")[0], synth.mutate(base, np.random.default_rng(L + 1), "ont")[0], 0))
    base = rng.integers(0, 4, 7000, dtype=np.uint8)
    clean = base.copy()
    noisy = synth.mutate(base, np.random.default_rng(9), "ont")[0]
    cases.append((clean, noisy, 0))                       # one side exact: long diagonal runs on the other side's errors only
    cases.append((noisy, clean, 0))
    exact = np.concatenate([synth.mutate(base[:2500], np.random.default_rng(10), "clr")[0], base[2500:4700], synth.mutate(base[4700:], np.random.default_rng(11), "clr")[0]])
    cases.append((exact, base, 0))                        # a 2,200-base exact run: the front jumps past a whole chunk in one step
    cases.append((base, exact, 0))
    hq = synth.mutate(base, np.random.default_rng(12), "hifi")[0]
    cases.append((hq, base, 1))                           # align_hq: narrow band, long runs
    cases.append((base[37:], synth.mutate(base, np.random.default_rng(13), "ont")[0][:6900], 0))
    for i, (q, t, is_hq) in enumerate(cases):
        qa, ta = util.ASC[q].tobytes(), util.ASC[t].tobytes()
        o, ots, oqs, _ = util.oracle_align(oracle_lib, qa, ta, is_hq)
        n, tu, qu, ts, qs = util.gpu_align(simt_lib, qa, ta, is_hq)
        assert n == o.aln_len, i
        if o.status == 1:
            assert ts == ots and qs == oqs and (tu, qu) == (o.t_used, o.q_used), i
    assert sum(1 for q, t, h in cases if len(q) > 2048) >= 10


@pytest.mark.parametrize("schedule", [0, 1, 2, 7])
def test_golden_piles_nextcorrect(simt_lib, schedule):
    """nextCorrect() end to end (K7 ... K11, the three-wave scoring pipeline with its default segments, the segmented walk)
    on the reference's golden piles.  Which wavefront of a workgroup runs when is not defined: schedule 0 gives every
    wavefront a slice in turn, 1 / 2 let the lowest / highest numbered runnable wavefront run ahead to its next workgroup
    barrier, larger values pick at random -- a missing barrier between wavefronts (the fault that cost round 2 thirty
    GPU-minutes, in the stitch kernel's segment loop) passes under 0 and 2 and fails under 1 and the random ones."""
    fn, fr = util.bind_correct(simt_lib)
    simt_lib.simt_set_schedule(schedule)
    simt_lib.simt_set_lane_order(1 if schedule in (2, 7) else 0)   # lanes of a wavefront highest first: see simt_runtime.cpp
    for i, p in enumerate(util.load_piles()):
        if schedule and i % 3 != schedule % 3:
            continue
        ln, ide, seq = util.call_correct(fn, fr, p)
        assert ln == p["exp_len"], i
        if ln > 4:
            assert seq == p["exp_seq"], i
            assert np.float32(ide) == np.float32(p["exp_ide"]), i
    simt_lib.simt_set_schedule(0)
    simt_lib.simt_set_lane_order(0)


_CHILD = r"""
import ctypes as C, json, sys
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, util, build_simt
from nextdenovo_amd import api
lib = api._bind(C.CDLL(build_simt.build()))
fn, fr = util.bind_correct(lib)
bad = []
for i, p in enumerate(util.load_piles()):
    if i %% %d:
        continue
    ln, ide, seq = util.call_correct(fn, fr, p)
    if ln != p["exp_len"] or (ln > 4 and (seq != p["exp_seq"] or np.float32(ide) != np.float32(p["exp_ide"]))):
        bad.append(i)
st = api.Stats()
lib.ndgpu_get_stats(C.byref(st))
print(json.dumps(dict(bad=bad, segments=int(st.score_segments), repairs=int(st.score_repairs), slow=int(st.score_slow_piles),
                      lq_rounds=int(st.lq_rounds), lq_declined=int(st.lq_declined), lq_jobs=int(st.lq_jobs),
                      lq_repairs=int(st.lq_repairs), tb_tasks=int(st.tb_tasks), tb_walkers=int(st.tb_walkers),
                      tb_fallbacks=int(st.tb_fallbacks))))
"""


def _forced(env, stride=2):
    e = dict(os.environ, NDGPU_CONTEXTS="1", **env)
    out = subprocess.run([sys.executable, "-c", _CHILD % (os.path.dirname(HERE), HERE, os.path.join(HERE, "simt"), stride)], env=e,
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("env,expect", [
    ({"NDGPU_K10_SEG": "256", "NDGPU_K10_WARM": "3"}, "repairs"),       # warm-up too short: boundary checks fail, in-kernel repair
    ({"NDGPU_K10_FORCE": "repair", "NDGPU_K10_SEG": "512", "SIMT_SCHEDULE": "1"}, "repairs"),  # every other segment repaired
    ({"NDGPU_K10_FORCE": "repair", "NDGPU_K10_SEG": "300", "SIMT_SCHEDULE": "9"}, "repairs"),
    ({"NDGPU_K10_FORCE": "slow"}, "slow"),                               # the int64 HBM-resident kernel
    ({"NDGPU_K10_FORCE": "large", "NDGPU_K10_SEG": "300", "NDGPU_K10_WARM": "64"}, "segments"),
    ({"SIMT_LANES_DESCENDING": "1", "SIMT_SCHEDULE": "1"}, "segments"),  # lane order / schedule must not matter to the register-path K7
])
def test_scoring_forced_paths(simt_lib, env, expect):
    """The scoring kernel's other paths (see tests/test_gpu_k10.py for the same switches on the GPU)."""
    r = _forced(env)
    assert r["bad"] == [], r
    assert r[expect] > 0, r


def test_lq_rounds_on_the_device_and_on_the_host(simt_lib):
    """The low-quality-region rounds: K12 takes every round of the golden piles (none declined); with NDGPU_LQ_HOST the same
    rounds go the host way (alignments as a batch, second MSA in the engine) -- both give the reference's records."""
    dev = _forced({}, stride=1)
    assert dev["bad"] == [] and dev["lq_rounds"] >= 10 and dev["lq_declined"] == 0, dev
    host = _forced({"NDGPU_LQ_HOST": "1", "SIMT_SCHEDULE": "1"}, stride=1)
    assert host["bad"] == [] and host["lq_rounds"] == 0, host
    # piles whose low-quality regions add up to more columns than K12 takes go the host way, the others stay on the device
    mixed = _forced({"NDGPU_K12_MAX_COLUMNS": "150"}, stride=1)
    assert mixed["bad"] == [] and 0 < mixed["lq_declined"] < mixed["lq_rounds"], mixed


@pytest.mark.parametrize("env,repairs", [
    ({"NDGPU_K12_JOB_COLUMNS": "1"}, False),                                  # every region a job: a speculative start per region
    ({"NDGPU_K12_JOB_COLUMNS": "1", "NDGPU_K12_WARM": "1"}, None),            # one warm-up column: boundary checks fail and are repaired
    ({"NDGPU_K12_JOB_COLUMNS": "1", "NDGPU_K12_WARM": "3", "SIMT_LANES_DESCENDING": "1"}, None),
    ({"NDGPU_K12_JOB_COLUMNS": "1", "NDGPU_K12_FORCE": "repair"}, True),      # every second job scored again by the stitch kernel
    ({"NDGPU_K12_JOB_COLUMNS": "40", "NDGPU_K12_FORCE": "repair", "NDGPU_K12_WARM": "16"}, True),
])
def test_lq_rounds_cut_into_speculative_jobs(simt_lib, env, repairs):
    """K12b scores a pile's low-quality-region MSA job by job from speculative starts (a warm-up on the tail of the job before, every
    unknown score one constant), the stitch kernel checks every boundary and scores a job that fails again from the true scores, the
    walk is cut at the 'N' cells: whatever the cut and the warm-up, the records are the reference's."""
    r = _forced(env, stride=1)
    assert r["bad"] == [] and r["lq_declined"] == 0, r
    assert r["lq_jobs"] > r["lq_rounds"], r          # piles were cut into several jobs
    if repairs is True:
        assert r["lq_repairs"] > 0, r
    elif repairs is False:
        assert r["lq_repairs"] == 0, r


@pytest.mark.parametrize("env,fallbacks", [
    ({"NDGPU_K8_SEG": "16", "NDGPU_K8_WARM": "8", "NDGPU_K8_MINLEN": "0"}, None),     # ~60 walkers per kb of alignment, main phase and rounds alike
    ({"NDGPU_K8_SEG": "16", "NDGPU_K8_WARM": "1", "NDGPU_K8_MINLEN": "0", "SIMT_SCHEDULE": "1"}, True),   # one warm-up row: boundaries disagree, the one-lane walk takes over
    ({"NDGPU_K8_SEG": "64", "NDGPU_K8_WARM": "64", "NDGPU_K8_MINLEN": "0", "SIMT_LANES_DESCENDING": "1"}, None),
    ({"NDGPU_K8_MINLEN": "0"}, False),                                                  # the product's segment length and warm-up on every launch
    ({"NDGPU_K8_SEG": "0"}, "off"),                                                     # the one-lane kernel everywhere
    ({"NDGPU_K8_SEG": "16", "NDGPU_K8_WARM": "8", "NDGPU_K8_MINLEN": "0", "NDGPU_TB_WIN": "0"}, None),   # walkers that read memory directly
])
def test_traceback_in_segments(simt_lib, env, fallbacks):
    """K8a cut into walkers (checkpoints of the forward kernel, chase, one lane per 2^k edit steps from a speculative x, stitch, the
    one-lane walk for what the stitch refuses): whatever the segment length and the warm-up, the records are the reference's."""
    r = _forced(env, stride=1)
    assert r["bad"] == [], r
    if fallbacks == "off":
        assert r["tb_tasks"] == 0, r
        return
    assert r["tb_tasks"] > 100 and r["tb_walkers"] > r["tb_tasks"], r
    if fallbacks is True:
        assert r["tb_fallbacks"] > 0, r
    elif fallbacks is False:
        assert r["tb_fallbacks"] == 0, r


_CHILD_ALIGN = r"""
import ctypes as C, json, sys
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, util, build_simt
from nextdenovo_amd import api, synth
lib = C.CDLL(build_simt.build())
api._LIB = api._bind(lib)
ora = C.CDLL(%r)
rng = np.random.default_rng(11)
bad, n_ok = [], 0
for p in util.load_pairs():
    n, tu, qu, ts, qs = util.gpu_align(lib, p["q"], p["t"], p["hq"])
    if n != p["aln_len"] or (n > 2 and (not np.array_equal(util.strings_to_ops(ts, qs), p["ops"]) or (tu, qu) != (p["t_used"], p["q_used"]))):
        bad.append("golden")
for it in range(60):
    L = int(rng.integers(50, 5000))
    kind = it %% 6
    base = rng.integers(0, 4 if kind else 2, L, dtype=np.uint8)
    if kind == 3 and L > 200:                       # homopolymers and tandem repeats: the walk's overshoot can outlast many rows
        for _ in range(20):
            unit = rng.integers(0, 4, int(rng.integers(1, 4)), dtype=np.uint8)
            at = int(rng.integers(0, L - 90))
            base[at:at + 80] = np.resize(unit, 80)
    prof = ("ont", "clr", "hifi")[it %% 3]
    q = synth.mutate(base, np.random.default_rng(3 * it + 100), prof)[0]
    t = synth.mutate(base, np.random.default_rng(3 * it + 101), prof)[0]
    if kind == 4:                                   # a query prefix the target lacks: the walk runs along the edge (forced moves, lib/align.c:512)
        q = np.concatenate([rng.integers(0, 4, int(rng.integers(5, 150)), dtype=np.uint8), q])
    if kind == 5:
        t = np.concatenate([rng.integers(0, 4, int(rng.integers(5, 150)), dtype=np.uint8), t])
    hq = int(it %% 7 == 0)
    qa, ta = util.ASC[q].tobytes(), util.ASC[t].tobytes()
    o, ots, oqs, _ = util.oracle_align(ora, qa, ta, hq)
    n, tu, qu, ts, qs = util.gpu_align(lib, qa, ta, hq)
    if n != o.aln_len or (o.status == 1 and (ts != ots or qs != oqs or (tu, qu) != (o.t_used, o.q_used))):
        bad.append(it)
    n_ok += o.status == 1
st = api.stats()
print(json.dumps(dict(bad=bad, ok=int(n_ok), tb_tasks=st["tb_tasks"], tb_walkers=st["tb_walkers"], tb_fallbacks=st["tb_fallbacks"])))
"""


@pytest.mark.parametrize("env", [
    {"NDGPU_K8_SEG": "16", "NDGPU_K8_WARM": "4", "NDGPU_K8_MINLEN": "0"},
    {"NDGPU_K8_SEG": "32", "NDGPU_K8_WARM": "32", "NDGPU_K8_MINLEN": "0", "SIMT_LANES_DESCENDING": "1"},
    {"NDGPU_K8_SEG": "256", "NDGPU_K8_WARM": "32", "NDGPU_K8_MINLEN": "0"},
])
def test_align_in_segments_vs_oracle(simt_lib, oracle_lib, env):
    """The C-ABI `align` with the traceback in segments: the reference's golden pairs (failures, the > 250-gap marker, wide bands) and 60
    fuzzed pairs -- low-complexity sequence, one-sided overhangs -- against the oracle, with boundaries that do and do not agree."""
    e = dict(os.environ, NDGPU_CONTEXTS="1", **env)
    out = subprocess.run([sys.executable, "-c", _CHILD_ALIGN % (os.path.dirname(HERE), HERE, os.path.join(HERE, "simt"), oracle_lib._name)], env=e,
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["bad"] == [] and r["ok"] > 40, r
    assert r["tb_tasks"] > 60 and r["tb_walkers"] > r["tb_tasks"], r
    if env["NDGPU_K8_WARM"] == "4":
        assert r["tb_fallbacks"] > 0, r


def _synth_set(gsize, mu, sigma, seed, depth=30, profile="ont"):
    from nextdenovo_amd import synth
    g = synth.make_genome(gsize, seed=seed, n_repeats=0)
    rs = synth.simulate_reads(g, depth, profile, seed=seed + 1, mu=mu, sigma=sigma)
    return rs, synth.build_piles(rs, seed_cutoff=1000)


@pytest.mark.parametrize("profile,read_type,max_lq", [("ont", 1, 10000), ("hifi", 3, 1000)])
def test_db_path_equals_ascii_path_and_host_oracle(simt_api, host_harness, profile, read_type, max_lq):
    """Resident-DB batched entry (sub-batches, both strands, LQ-stage alignments) == per-pile ASCII entry == the host engine with
    the oracle's aligner, for raw and for high-quality reads."""
    from nextdenovo_amd import synth
    rs, piles = _synth_set(20000, 7.9, 0.3, 31 if read_type == 1 else 71, profile=profile)
    piles = piles[:6]
    words, off, lens = synth.pack_db(rs)
    db = simt_api.ReadDB(words, off, lens)
    recs, poff = synth.flatten_piles(piles)
    got = db.correct_piles(recs, poff, read_type=read_type, max_lq_length=max_lq, host_threads=4)
    db.close()
    fn, fr = util.bind_correct(host_harness, "ndtest_correct", "ndtest_free")
    assert any(int(p["recs"][:, 1].max()) == 1 for p in piles)
    for p, g in zip(piles, got):
        seqs, st, en, mal = synth.pile_sequences(rs, p)
        mlq = min(en[0] // 2, max_lq)
        c = util.call_correct(fn, fr, dict(seqs=seqs, aln_start=st, aln_end=en, max_aln=mal, max_lq=mlq, read_type=read_type,
                                           fast=0, split=0))
        assert g[0] == c[0] and g[0] > 1000
        assert g[2] == c[2]
        assert np.float32(g[1]) == np.float32(c[1])


_OOM_CHILD = r"""
import ctypes as C, json, sys
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, build_simt, chain_util
from nextdenovo_amd import api, synth
api._LIB = api._bind(C.CDLL(build_simt.build()))
g = synth.make_genome(20000, seed=31, n_repeats=0)
rs = synth.simulate_reads(g, 30, "ont", seed=32, mu=7.9, sigma=0.3)
piles = synth.build_piles(rs, seed_cutoff=1000)[:8]
words, off, lens = synth.pack_db(rs)
recs, poff = synth.flatten_piles(piles)
db = api.ReadDB(words, off, lens)
res = db.correct_piles(recs, poff, host_threads=4)
again = db.correct_piles(recs, poff, host_threads=4) if %d else res   # a second call after the failures: nothing is poisoned
db.close()
print(json.dumps(dict(a=[chain_util.digest(r) for r in res], b=[chain_util.digest(r) for r in again])))
"""


def _oom_child(env, twice=0):
    e = dict(os.environ, NDGPU_TRACE="1", NDGPU_CONTEXTS="1", NDGPU_SUBBATCHES_PER_CONTEXT="1", **env)
    out = subprocess.run([sys.executable, "-c", _OOM_CHILD % (os.path.dirname(HERE), HERE, os.path.join(HERE, "simt"), twice)], env=e,
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1]), out.stderr


def test_deep_pile_through_every_read_chunk_of_the_link_counter(simt_api, host_harness):
    """A pile of > 128 accepted reads: K9's first 64 reads keep a 32-byte window of their tag stream in registers, the second 64 take their
    next tag straight from the stream (round 6: that window was what the register budget spilled), the rest go through the column index
    -- all three against the host engine + oracle."""
    from nextdenovo_amd import synth
    rs, piles = _synth_set(6000, 7.6, 0.25, 83, depth=150)
    p = max(piles, key=lambda q: len(q["recs"]))
    assert len(p["recs"]) > 135, len(p["recs"])
    seqs, st, en, mal = synth.pile_sequences(rs, p)
    mlq = min(en[0] // 2, 10000)
    fn, fr = util.bind_correct(host_harness, "ndtest_correct", "ndtest_free")
    c = util.call_correct(fn, fr, dict(seqs=seqs, aln_start=st, aln_end=en, max_aln=mal, max_lq=mlq, read_type=1, fast=0, split=0))
    a = simt_api.correct(seqs, st, en, mal, max_lq_length=mlq)
    assert a[0] == c[0] and a[2] == c[2] and a[0] > 1000
    assert np.float32(a[1]) == np.float32(c[1])


def test_out_of_device_memory_halves_the_sub_batch(simt_lib):
    """lib/nextcorrect.c:2254-2261, lib/nextcorrect.py:255-257: a seed whose working memory cannot be had is a `len == 3` seed and
    nothing else is lost.  Here: a sub-batch that does not fit is halved until its pieces fit -- the records of the pieces are the
    records of the undisturbed run -- and only a single pile that does not fit alone is reported.  (Round 3 shipped a size mark that
    made every retry ask for the failed size again: every pile of the sub-batch came back as len 3.)"""
    want, _ = _oom_child({})
    assert len(want["a"]) == 8 and all(d[0] > 1000 for d in want["a"])
    # find a limit the whole sub-batch does not fit under but single piles do: the largest allocation of the undisturbed run / 3
    _, err = _oom_child({"NDGPU_DEBUG_ALLOC": "1"})
    big = max(int(m) for m in re.findall(r"\((\d+) bytes, asked", err))
    got, err = _oom_child({"NDGPU_OOM_ABOVE": str(big // 3)}, twice=1)
    assert "out of device memory" in err and "halved" in err
    assert got["a"] == want["a"], "halved sub-batches must give the undisturbed records"
    assert got["b"] == want["a"], "a later call must not inherit sizes that failed"
    # pinned host arenas fail the same way
    got, err = _oom_child({"NDGPU_PINNED_OOM_ABOVE": str(200 << 10)})
    assert "halved" in err and all(a == b or a[0] == 3 for a, b in zip(got["a"], want["a"]))
    # nothing fits: every seed is an out-of-memory seed, nothing aborts
    got, err = _oom_child({"NDGPU_OOM_ABOVE": str(64 << 10)})
    assert all(d[0] == 3 for d in got["a"]) and "out-of-memory seed" in err


def test_records_are_handed_over_as_their_sub_batches_finish(simt_api, tmp_path, monkeypatch):
    """ndgpu_correct_piles_stream: the completion callback reports every pile exactly once (a sub-batch at a time), and the cns.fasta /
    .idx written from it hold the records of the plain call, line for line as lib/nextcorrect.py:236-260 prints them."""
    from nextdenovo_amd import synth
    monkeypatch.setenv("NDGPU_SUBBATCH", "2")   # several sub-batches, two contexts: callbacks from more than one thread
    monkeypatch.setenv("NDGPU_CONTEXTS", "2")
    rs, piles = _synth_set(20000, 7.9, 0.3, 31)
    piles = piles[:7]
    words, off, lens = synth.pack_db(rs)
    db = simt_api.ReadDB(words, off, lens)
    recs, poff = synth.flatten_piles(piles)
    names = [int(p["seed"]) for p in piles]
    plain = db.correct_piles(recs, poff, read_type=1, max_lq_length=10000, host_threads=4)
    fa = tmp_path / "cns.fasta"
    with open(fa, "wb") as OUT, open(str(fa) + ".idx", "wb") as IDX:
        OUT.write(b">0 5 1.000000\nACGTA\n")      # (the files may hold records already: offsets count from the file's start)
        res = db.correct_piles(recs, poff, read_type=1, max_lq_length=10000, host_threads=4, fasta=(OUT, IDX, names, 500, 0.8))
        assert OUT.tell() == os.path.getsize(fa) and IDX.tell() == os.path.getsize(str(fa) + ".idx")
    # the same through objects that are not files (the record-by-record loop instead of ndgpu_write_records): the same records
    import io
    mem_out, mem_idx = io.BytesIO(), io.BytesIO()
    mem_out.write(b">0 5 1.000000\nACGTA\n")
    res2 = db.correct_piles(recs, poff, read_type=1, max_lq_length=10000, host_threads=4, fasta=(mem_out, mem_idx, names, 500, 0.8))
    db.close()
    assert res2 == res

    def records(b):
        return sorted(b.split(b">")[1:])
    assert records(mem_out.getvalue()) == records(open(fa, "rb").read())
    assert sorted(ln.split()[0::2] for ln in mem_idx.getvalue().splitlines()) == sorted(ln.split()[0::2] for ln in open(str(fa) + ".idx", "rb").read().splitlines())
    assert [(ln, np.float32(ide)) for ln, ide in res] == [(ln, np.float32(ide)) for ln, ide, _ in plain]
    blob = open(fa, "rb").read()
    got = {}
    at = len(b">0 5 1.000000\nACGTA\n")
    assert blob[:at] == b">0 5 1.000000\nACGTA\n"
    while at < len(blob):
        e1 = blob.index(b"\n", at)
        e2 = blob.index(b"\n", e1 + 1)
        name, ln, ide = blob[at + 1:e1].split()
        got[int(name)] = (int(ln), ide.decode(), blob[e1 + 1:e2], e1 + 1)
        at = e2 + 1
    want = {n: (ln, "%f" % ide, seq) for n, (ln, ide, seq) in zip(names, plain) if ln >= 500 and ide >= 0.8}
    assert len(want) >= 5 and {n: v[:3] for n, v in got.items()} == want
    idx = dict((int(a), (int(b), int(c))) for a, b, c in (ln.split() for ln in open(str(fa) + ".idx")))
    assert set(idx) == set(names)
    for n, (ln, _ide, _seq, pos) in got.items():
        assert idx[n] == (pos, ln)                 # offset of the record's first base, its length (lib/nextcorrect.py:249-251)
    for n in set(names) - set(got):
        assert idx[n] == (0, 0)
