"""N > 1 path of bench.py on CPU: world_size 2 over gloo.  Each rank builds its own read
set (weak scaling: one read set per rank, piles never cross ranks), corrects it with the
host engine + oracle backend, and the ranks meet only in the final sum/max reduction."""
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


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import ctypes as C

    import torch
    import torch.distributed as dist

    import bench
    import util
    from nextdenovo_amd import synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    h = C.CDLL(os.path.join(HERE, "csrc", "libndhost_test.so"))
    fn, fr = util.bind_correct(h, "ndtest_correct", "ndtest_free")
    g = synth.make_genome(12000, seed=42 + 1000 * rank, n_repeats=0)
    rs = synth.simulate_reads(g, 25, "ont", seed=43 + 1000 * rank, mu=7.8, sigma=0.3)
    piles = synth.build_piles(rs, seed_cutoff=1000)[:3]
    bases = 0
    for p in piles:
        seqs, st, en, mal = synth.pile_sequences(rs, p)
        ln, ide, _ = util.call_correct(fn, fr, dict(seqs=seqs, aln_start=st, aln_end=en, max_aln=mal,
                                                    max_lq=min(en[0] // 2, 10000), read_type=1, fast=0, split=0))
        if ln > 4 and ide >= 0.8:
            bases += ln
    total, tmax = bench.reduce_over_ranks(dist, torch, bases, 1.0 + rank, "cpu")
    q.put((rank, bases, total, tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduce(host_harness):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    (r0, b0, t0, m0), (r1, b1, t1, m1) = res
    assert b0 > 0 and b1 > 0 and b0 != b1          # different read sets per rank
    assert t0 == t1 == b0 + b1                      # sum over ranks
    assert m0 == m1 == 2.0                          # max over ranks
