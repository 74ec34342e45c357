"""Scoring DP (K10) at BASELINE config-2 size: the segment-parallel kernel against the compiled reference, and every one
of its paths -- large tables, int64 HBM-resident kernel, unsegmented, repaired segments (forced and natural), the raw-score
guard -- against each other on the same piles.  lib/nextcorrect.c:2149-2202 is what all of them compute."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import chain_util
import refpipe
import util

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GENOME, DEPTH = 4.6e6, 50.0   # BASELINE.json configs[1]


def _driver(env_extra, genome=GENOME, depth=DEPTH, max_piles=0, timeout=1500):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(HERE, "chain_util.py"), str(genome), str(depth), str(max_piles)],
                       capture_output=True, text=True, env=env, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.fixture(scope="module")
def default_run():
    return _driver({})


def _ref_worker(item):
    import ctypes as C
    import hashlib
    import struct
    seqs, st, en, mal, mlq = item
    lib = C.CDLL(os.path.join(refpipe.REFDIR, "nextcorrect.so"))
    fn, fr = util.bind_correct(lib)
    ln, ide, seq = util.call_correct(fn, fr, dict(seqs=seqs, aln_start=st, aln_end=en, max_aln=mal, max_lq=mlq, read_type=1,
                                                  fast=0, split=0))
    if ln <= 4:
        return (int(ln), 0, "")
    return (int(ln), struct.unpack("<I", struct.pack("<f", ide))[0], hashlib.md5(seq).hexdigest())


@pytest.mark.skipif(not refpipe.have_ref("nextcorrect.so"), reason="compiled reference did not travel")
def test_config2_piles_match_compiled_reference(default_run):
    """>= 300 piles of the config-2 chain incl. every seed >= 100 kb: len, identity bits, md5 == oracle/_ref/nextcorrect.so."""
    from multiprocessing import get_context
    from nextdenovo_amd import synth
    rs = chain_util.make_set(GENOME, DEPTH)
    sub, off, seeds, _n_bl, _db = chain_util.device_piles(rs)
    assert [int(s) for s in seeds] == default_run["seeds"]
    slen = sub[off[:-1].astype(np.int64), 3].astype(np.int64) + 1
    assert slen.max() >= 150000
    pick = sorted(set(np.nonzero(slen >= 100000)[0].tolist()) | set(range(0, seeds.size, max(1, seeds.size // 300))))
    assert len(pick) >= 300 and (slen[pick] >= 100000).sum() >= 10
    items = []
    for i in pick:
        p = {"seed": int(seeds[i]), "recs": sub[int(off[i]):int(off[i + 1])]}
        seqs, st, en, mal = synth.pile_sequences(rs, p)
        items.append((seqs, st, en, mal, min(en[0] // 2, 10000)))
    order = sorted(range(len(items)), key=lambda j: -len(items[j][0][0]))  # longest first: they bound the pool's wall time
    with get_context("fork").Pool(min(os.cpu_count() or 1, 64)) as pool:
        got = pool.map(_ref_worker, [items[j] for j in order], chunksize=1)
    want = {pick[j]: g for j, g in zip(order, got)}
    bad = [i for i in pick if tuple(default_run["digests"][i]) != want[i]]
    assert not bad, (len(bad), bad[:5])
    assert default_run["stats"]["score_segments"] > 50000 and default_run["stats"]["score_slow_piles"] == 0


@pytest.mark.parametrize("env,check", [
    ({"NDGPU_K10_FORCE": "seq"}, lambda s: s["score_segments"] == s["piles"]),
    ({"NDGPU_K10_FORCE": "large"}, lambda s: s["score_slow_piles"] == 0),
    ({"NDGPU_K10_FORCE": "repair"}, lambda s: s["score_repairs"] > 20000),
    ({"NDGPU_K10_WARM": "3", "NDGPU_K10_SEG": "256"}, lambda s: s["score_repairs"] > 100),   # warm-up too short: real failed checks
    ({"NDGPU_K10_SEG": "4096", "NDGPU_K10_WARM": "512"}, lambda s: s["score_slow_piles"] == 0),
    ({"NDGPU_K10_GUARD": str((1 << 29) + 200000)}, lambda s: s["score_slow_piles"] > 0),    # the int32 working range trips -> int64 kernel
], ids=["unsegmented", "large-tables", "forced-repair", "short-warmup", "long-segments", "guard"])
def test_scoring_paths_agree(default_run, env, check):
    got = _driver(env)
    assert got["seeds"] == default_run["seeds"]
    bad = [i for i, (a, b) in enumerate(zip(got["digests"], default_run["digests"])) if a != b]
    assert not bad, (len(bad), bad[:5])
    assert check(got["stats"]), got["stats"]


@pytest.mark.parametrize("env,check", [
    ({"NDGPU_K8_SEG": "0"}, lambda s: s["tb_tasks"] == 0),                                              # the one-lane kernel everywhere
    ({"NDGPU_K8_MINLEN": "0"}, lambda s: s["tb_fallbacks"] * 1000 <= s["tb_tasks"]),                    # the rounds' short alignments in segments too
    ({"NDGPU_K8_SEG": "64", "NDGPU_K8_WARM": "2"}, lambda s: s["tb_fallbacks"] > 100),                  # warm-up too short: the stitch refuses, the one-lane walk takes over
    ({"NDGPU_K8_SEG": "1024", "NDGPU_K8_WARM": "64"}, lambda s: s["tb_fallbacks"] * 1000 <= s["tb_tasks"]),
    ({"NDGPU_TB_WIN": "0"}, lambda s: s["tb_fallbacks"] * 1000 <= s["tb_tasks"]),                      # walkers without their LDS windows
], ids=["one-lane", "every-launch", "short-warmup", "long-segments", "no-windows"])
def test_traceback_forms_agree(default_run, env, check):
    """K8a in segments (the default: 256 rows a walker, 32 rows of warm-up) against the one-lane walk and against other cuts, on every
    config-2 pile; the default run itself is held against the compiled reference above."""
    st = default_run["stats"]
    assert st["tb_tasks"] > 100000 and st["tb_walkers"] > 5 * st["tb_tasks"] and st["tb_fallbacks"] * 1000 <= st["tb_tasks"], st
    got = _driver(env)
    assert got["seeds"] == default_run["seeds"]
    bad = [i for i, (a, b) in enumerate(zip(got["digests"], default_run["digests"])) if a != b]
    assert not bad, (len(bad), bad[:5])
    assert check(got["stats"]), got["stats"]


def test_lq_rounds_host_path_agrees(default_run):
    """The low-quality-region rounds on the device (K12: every round of the config-2 piles, a handful declined at most) against
    the host path of the same rounds (NDGPU_LQ_HOST: alignments as a batch, second MSA in the engine)."""
    st = default_run["stats"]
    assert st["lq_rounds"] >= 2 * 0.9 * st["piles"] and st["lq_declined"] <= st["lq_rounds"] // 100, st
    host = _driver({"NDGPU_LQ_HOST": "1"})
    assert host["stats"]["lq_rounds"] == 0
    assert host["seeds"] == default_run["seeds"]
    bad = [i for i, (a, b) in enumerate(zip(host["digests"], default_run["digests"])) if a != b]
    assert not bad, (len(bad), bad[:5])


def test_int64_kernel_agrees(default_run):
    """score_slow (every table in HBM, int64) on an even sample of the same piles incl. the longest seeds."""
    a = _driver({"NDGPU_K10_FORCE": "slow"}, max_piles=160)
    b = _driver({}, max_piles=160)
    assert a["seeds"] == b["seeds"] and a["digests"] == b["digests"]
    assert a["stats"]["score_slow_piles"] == a["stats"]["piles"] and b["stats"]["score_slow_piles"] == 0
    pos = {s: i for i, s in enumerate(default_run["seeds"])}
    assert all(tuple(d) == tuple(default_run["digests"][pos[s]]) for s, d in zip(a["seeds"], a["digests"]))


def test_sharded_stage_on_device_equals_oracle_backend(oracle_lib):
    """nextdenovo_amd.stage.Shard (what bench.py --gpus N gives every rank) with three seed files: the device backend's piles
    == the oracle backend's (overlap oracle + sort oracle, both pinned to the compiled reference), seed file by seed file."""
    import stage_util
    from nextdenovo_amd import stage, synth
    g = synth.make_genome(60000, seed=5, n_repeats=0)
    rs = synth.simulate_reads(g, 25, "ont", seed=6, mu=8.2, sigma=0.4)
    words, word_off, lens = synth.pack_db(rs)
    dev = stage.Shard(words, word_off, lens, n_seed_files=3, sort_k=20)
    ora = stage.Shard(words, word_off, lens, n_seed_files=3, sort_k=20, backend=stage_util.OracleBackend(oracle_lib, "ava-ont"))
    n_piles = 0
    for i in range(3):
        a, b = dev.piles(i), ora.piles(i)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]
        n_piles += a[2].size
    assert n_piles >= 10
