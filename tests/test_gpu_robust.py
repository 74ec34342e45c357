"""Failure handling of the consensus entry points on the device: out of device memory -> the sub-batch is halved, a pile that
does not fit alone is the reference's out-of-memory seed (len 3, lib/nextcorrect.c:2254-2261, lib/nextcorrect.py:255-257);
records that do not belong to the DB are refused; several DB handles can be alive."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import chain_util

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _driver(env_extra, genome=300000, depth=30):
    env = dict(os.environ, NDGPU_TRACE="1", **env_extra)
    r = subprocess.run([sys.executable, os.path.join(HERE, "chain_util.py"), str(genome), str(depth)], capture_output=True,
                       text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), r.stderr


def test_out_of_device_memory_is_survived():
    want, _ = _driver({})
    assert len(want["digests"]) > 50 and all(d[0] > 4 for d in want["digests"])
    # one sub-batch of all piles, allocations above 100 MB fail: it does not fit, its halves / quarters / ... do -> identical records
    got, err = _driver({"NDGPU_OOM_ABOVE": str(100 << 20), "NDGPU_CONTEXTS": "1", "NDGPU_SUBBATCHES_PER_CONTEXT": "1"})
    assert "out of device memory" in err and "halved" in err
    same = sum(1 for a, b in zip(got["digests"], want["digests"]) if a == b)
    oom = sum(1 for a in got["digests"] if a[0] == 3)
    assert same + oom == len(want["digests"]) and same >= len(want["digests"]) // 2, (same, oom)
    # allocations above 64 KB fail: no pile fits -> every seed comes back as len 3, nothing aborts
    got, err = _driver({"NDGPU_OOM_ABOVE": str(64 << 10)})
    assert all(d[0] == 3 for d in got["digests"]) and "out-of-memory seed" in err


def test_foreign_records_are_refused_and_handles_are_independent():
    from nextdenovo_amd import api, synth
    rs_a = chain_util.make_set(120000, 25, seed=3, mu=8.5, sigma=0.4)
    rs_b = chain_util.make_set(90000, 25, seed=9, mu=8.5, sigma=0.4)
    pa = synth.build_piles(rs_a, seed_cutoff=1000)[:12]
    pb = synth.build_piles(rs_b, seed_cutoff=1000)[:12]
    ra, oa = synth.flatten_piles(pa)
    rb, ob = synth.flatten_piles(pb)
    db_a = api.ReadDB(*synth.pack_db(rs_a))
    alone = db_a.correct_piles(ra, oa)
    db_b = api.ReadDB(*synth.pack_db(rs_b))          # a second live handle must not disturb the first
    assert db_a.correct_piles(ra, oa) == alone
    got_b = db_b.correct_piles(rb, ob)
    assert db_a.correct_piles(ra, oa) == alone and db_b.correct_piles(rb, ob) == got_b
    assert all(r[0] > 1000 for r in alone) and all(r[0] > 1000 for r in got_b)
    bad = ra.copy()
    bad[3, 4] = len(rs_a) + 5                          # a read id beyond the DB
    with pytest.raises(ValueError):
        db_a.correct_piles(bad, oa)
    bad = ra.copy()
    bad[2, 6] = 10 ** 7                                # a window beyond its read
    with pytest.raises(ValueError):
        db_a.correct_piles(bad, oa)
    db_a.close()
    assert db_b.correct_piles(rb, ob) == got_b         # destroying one handle leaves the other's device copy alone
    db_b.close()
