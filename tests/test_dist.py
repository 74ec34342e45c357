"""N > 1 path of bench.py on CPU: world_size 2 over gloo, STRONG scaling.  ONE read set; its seeds are dealt round robin
into two seed files (util/seq_dump.c:87-92); rank r computes the raw_align jobs of seed file r (its own and the mirrors
it needs, nextDenovo:426-467), sorts them, assembles the piles and corrects them -- here with the oracle overlap / sort
backend and the host engine + oracle aligner, since there is no GPU.  The ranks meet only in the final reduction, and the
union of their records is the single-process run over both seed files."""
import hashlib
import os
import socket
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _correct_file(i, n_files, exchange_dir=None):
    """{seed id: (len, md5)} of seed file i of n_files of the shared read set, CPU backends.  With `exchange_dir` the ranks hand
    the seed x seed jobs over (stage.Exchange): every pair is mapped by one rank only."""
    import ctypes as C

    import numpy as np

    import stage_util
    import util
    from nextdenovo_amd import stage, synth
    olib = C.CDLL(os.path.join(ROOT, "oracle", "libndoracle.so"))
    h = C.CDLL(os.path.join(HERE, "csrc", "libndhost_test.so"))
    fn, fr = util.bind_correct(h, "ndtest_correct", "ndtest_free")
    g = synth.make_genome(30000, seed=42, n_repeats=0)
    rs = synth.simulate_reads(g, 22, "ont", seed=43, mu=8.0, sigma=0.35)
    words, word_off, lens = synth.pack_db(rs)
    sh = stage.Shard(words, word_off, lens, preset="ava-ont", seed_cutoff=1000, read_cutoff=500, n_seed_files=n_files, sort_k=17,
                     blacklist=False, backend=stage_util.OracleBackend(olib, "ava-ont"),
                     exchange=stage.Exchange(exchange_dir, i, timeout_s=300.0) if exchange_dir else None)
    sub, off, seeds, _n_bl = sh.piles(i)
    out = {}
    for p in range(seeds.size):
        pile = {"seed": int(seeds[p]), "recs": sub[int(off[p]):int(off[p + 1])]}
        seqs, st, en, mal = synth.pile_sequences(rs, pile)
        ln, ide, seq = util.call_correct(fn, fr, dict(seqs=seqs, aln_start=st, aln_end=en, max_aln=mal, max_lq=min(en[0] // 2, 10000),
                                                     read_type=1, fast=0, split=0))
        out[int(seeds[p])] = (int(ln), hashlib.md5(seq).hexdigest())
    if exchange_dir:
        _correct_file.exchange_stats = dict(sh.exchange.stats)
    return out, [int(x) for x in sh.seed_ids[i]], sh.jobs_of(i)


def _worker(rank, world, port, q, exchange_dir=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist

    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_files, mine = bench.shard_of_rank(world, rank, 0, 0)
    recs, seed_ids, jobs = _correct_file(mine, n_files, exchange_dir)
    bases = sum(ln for ln, _ in recs.values() if ln > 4)
    total, tmax = bench.reduce_over_ranks(dist, torch, bases, 1.0 + rank, "cpu", len(recs))
    per = bench.gather_rank_stats(dist, torch, "cpu", [1.0 + rank, 0.5, 0.25, 0.125, float(len(recs)), float(bases)])
    assert len(per) == world and per[rank]["piles"] == len(recs) and per[1 - rank]["wall_s"] == 2.0 - rank
    q.put((rank, recs, seed_ids, jobs, bases, total, tmax, bench.reduce_over_ranks.seeds) + ((_correct_file.exchange_stats,) if exchange_dir else ()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_share_one_read_set(host_harness, oracle_lib):
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, rec0, ids0, jobs0, b0, t0, m0, s0), (_, rec1, ids1, jobs1, b1, t1, m1, s1) = res
    # the two seed files partition the seeds, dealt alternately in read order
    assert not set(ids0) & set(ids1) and sorted(ids0 + ids1) == list(range(len(ids0) + len(ids1)))
    assert ids0 == list(range(0, len(ids0) + len(ids1), 2)) and ids1 == list(range(1, len(ids0) + len(ids1), 2))
    # rank 0 runs (0, seed 0) and (0, seed 1); rank 1 needs the mirror (0, seed 1) and its own (1, seed 1): job order kept
    assert [j[:4] for j in jobs0] == [(0, 0, "seed", 0), (1, 0, "seed", 1)]
    assert [j[:4] for j in jobs1] == [(1, 0, "seed", 1), (2, 1, "seed", 1)]
    assert b0 > 0 and b1 > 0 and t0 == t1 == b0 + b1 and m0 == m1 == 2.0 and s0 == s1 == len(rec0) + len(rec1)
    # union of the ranks' records == one process correcting both seed files
    single = {}
    for i in range(2):
        single.update(_correct_file(i, 2)[0])
    both = dict(rec0)
    both.update(rec1)
    assert both == single and len(single) >= 6 and sum(1 for ln, _ in single.values() if ln > 1000) >= 6


def test_two_ranks_hand_the_mirror_job_over(host_harness, oracle_lib, tmp_path):
    """nextDenovo:455-459: the reference maps a seed x seed pair once and links the file for its second reader.  With
    stage.Exchange rank 0 owns (0, seed 0) and (0, seed 1), writes the latter into the node-local directory, rank 1 computes only
    (1, seed 1) and reads the mirror: the records of both ranks are those of the run in which every rank maps everything it needs."""
    import torch.multiprocessing as mp

    from nextdenovo_amd import stage
    sys.path.insert(0, HERE)
    assert [stage.owner_of(0, 0), stage.owner_of(0, 1), stage.owner_of(1, 1)] == [0, 0, 1]
    pl = stage.plan(8)
    assert max(r["jobs_computed_with_exchange"] for r in pl) == 5 and sum(r["jobs_computed_with_exchange"] for r in pl) == 36
    assert all(r["jobs_computed_alone"] == 8 for r in pl)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    xdir = str(tmp_path / "xchg")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, xdir)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, rec0, *_r0, x0), (_, rec1, *_r1, x1) = res
    assert x0["sent"] == 1 and x0["received"] == 0 and x1["received"] == 1 and x1["sent"] == 0 and x0["recomputed"] == x1["recomputed"] == 0
    single = {}
    for i in range(2):
        single.update(_correct_file(i, 2)[0])
    both = dict(rec0)
    both.update(rec1)
    assert both == single and len(single) >= 6


def test_a_late_owner_does_not_stop_the_step(host_harness, oracle_lib, tmp_path):
    """The hand-over never blocks a step for good: a job whose owner does not deliver within the timeout is computed by the rank that
    needs it (same orientation, same records), and the count says so."""
    import ctypes as C

    import numpy as np

    import stage_util
    from nextdenovo_amd import stage, synth
    olib = C.CDLL(os.path.join(ROOT, "oracle", "libndoracle.so"))
    g = synth.make_genome(30000, seed=42, n_repeats=0)
    rs = synth.simulate_reads(g, 22, "ont", seed=43, mu=8.0, sigma=0.35)
    words, word_off, lens = synth.pack_db(rs)

    def shard(ex):
        return stage.Shard(words, word_off, lens, preset="ava-ont", seed_cutoff=1000, read_cutoff=500, n_seed_files=2, sort_k=17,
                           blacklist=False, backend=stage_util.OracleBackend(olib, "ava-ont"), exchange=ex)
    want = shard(None).piles(1)
    ex = stage.Exchange(str(tmp_path / "nobody_writes_here"), 1, timeout_s=0.3)   # rank 1 of 2: the mirror (0, seed 1) belongs to rank 0
    got = shard(ex).piles(1)
    assert ex.stats["recomputed"] == 1 and ex.stats["received"] == 0 and ex.stats["sent"] == 0
    assert all(np.array_equal(a, b) for a, b in zip(got[:3], want[:3])) and got[3] == want[3]


def test_the_owner_may_run_steps_ahead_of_the_reader(host_harness, oracle_lib, tmp_path):
    """Ranks are not synchronised between steps: the owner of a mirror job may be several steps ahead of the rank that reads it.  Files
    are kept until their one reader has read them (the reader removes them), so nothing is lost and nothing is recomputed."""
    import ctypes as C

    import numpy as np

    import stage_util
    from nextdenovo_amd import stage, synth
    olib = C.CDLL(os.path.join(ROOT, "oracle", "libndoracle.so"))
    g = synth.make_genome(24000, seed=42, n_repeats=0)
    rs = synth.simulate_reads(g, 20, "ont", seed=43, mu=8.0, sigma=0.35)
    words, word_off, lens = synth.pack_db(rs)
    xdir = str(tmp_path / "x")

    def shard(rank):
        return stage.Shard(words, word_off, lens, preset="ava-ont", seed_cutoff=1000, read_cutoff=500, n_seed_files=2, sort_k=17,
                           blacklist=False, backend=stage_util.OracleBackend(olib, "ava-ont"), exchange=stage.Exchange(xdir, rank, timeout_s=5.0))
    s0, s1 = shard(0), shard(1)
    first = [s0.overlaps(0) for _ in range(3)]          # rank 0 runs three steps: three files of the job (0, seed 1) wait
    assert s0.exchange.stats["sent"] == 3 and len(os.listdir(xdir)) == 3
    got = [s1.overlaps(1) for _ in range(3)]            # rank 1 catches up, step by step
    assert s1.exchange.stats == {"sent": 0, "received": 3, "recomputed": 0, "wait_s": s1.exchange.stats["wait_s"]}
    assert os.listdir(xdir) == []
    for step in range(3):                               # the mirror rank 1 read is the job rank 0 computed in that step
        assert np.array_equal(got[step][0], first[step][1])
