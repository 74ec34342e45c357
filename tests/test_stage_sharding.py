"""nextdenovo_amd.stage.Shard (what bench.py --gpus N gives every rank) against the reference PROGRAMS on the CPU: reads ->
compiled `seq_dump -n 2` (two seed files + a part file) -> compiled `minimap2-nd --step 1` for the five raw_align jobs of
nextDenovo:426-467 -> compiled `ovl_sort` per seed file -> the pile assembly of lib/nextcorrect.py:92-143 through the stock
ovlseq.so.  The shard runs with the oracle backend (overlap oracle + sort oracle, no GPU here): the records of every job it
computes for a seed file -- its own and the mirror it needs -- are the bytes of the reference's `.ovl` for that job, and the
piles it hands to the consensus are the reference's piles."""
import os

import numpy as np
import pytest

import mm_util as M
import os_util as O
import refpipe
import stage_util

NEED = ("minimap2-nd", "ovl_sort", "seq_dump", "ovlseq.so")


@pytest.mark.skipif(not refpipe.have_ref(*NEED), reason="compiled reference not built")
def test_shard_jobs_and_piles_equal_reference_programs(tmp_path, oracle_lib):
    from nextdenovo_amd import overlap, stage, synth
    g = synth.make_genome(45000, seed=141, n_repeats=2, repeat_len=1200)
    rs = synth.simulate_reads(g, 24, "ont", seed=142, mu=8.3, sigma=0.5, min_len=700)
    wd = str(tmp_path)
    fa = os.path.join(wd, "reads.fa")
    refpipe.write_fasta(fa, [synth.codes_to_ascii(s) for s in rs.seqs])
    fofn = os.path.join(wd, "input.fofn")
    open(fofn, "w").write(fa + "\n")
    db = os.path.join(wd, "db")
    os.makedirs(db)
    R = lambda n: os.path.join(refpipe.REFDIR, n)  # noqa: E731
    refpipe.run([R("seq_dump"), "-f", "500", "-s", "4000", "-b", "2g", "-n", "2", "-d", db, fofn])
    s1, s2, p1 = (os.path.join(db, n) for n in ("input.seed.001.2bit", "input.seed.002.2bit", "input.part.001.2bit"))
    assert os.path.getsize(s2) > 500 and os.path.getsize(p1) > 500
    ra = os.path.join(wd, "ra")
    os.makedirs(ra)
    jobs = [(0, s1, p1, True, None), (1, s1, s1, False, "3G"), (2, s1, s2, True, "3G"), (3, s2, p1, True, None), (4, s2, s2, False, "3G")]
    ref_ovl = {}
    for k, t, q, dual, batch in jobs:
        o = os.path.join(ra, "%s.%d.ovl" % (os.path.basename(t), k))
        refpipe.run([R("minimap2-nd"), "--step", "1"] + (["-I", batch] if batch else []) + (["--dual=yes"] if dual else []) +
                    ["-t", "4", "-x", "ava-ont", t, q, "-o", o])
        ref_ovl[k] = open(o, "rb").read()
        assert len(ref_ovl[k]) > 500
    words, word_off, lens = synth.pack_db(rs)   # read id = input order: every read is >= the 500-base read cut-off
    sh = stage.Shard(words, word_off, lens, preset="ava-ont", seed_cutoff=4000, read_cutoff=500, n_seed_files=2, sort_k=17,
                     backend=stage_util.OracleBackend(oracle_lib, "ava-ont"))
    assert len(sh.part_ids) == 1 and sh.part_ids[0].size > 3
    assert [j[0] for j in sh.jobs_of(0)] == [0, 1, 2] and [j[0] for j in sh.jobs_of(1)] == [2, 3, 4]
    idxs = os.path.join(wd, "idxs.fofn")
    with open(idxs, "w") as f:
        for n in sorted(os.listdir(db)):
            if n.startswith(".input.") and n.endswith(".idx"):
                f.write(os.path.join(db, n) + "\n")
    n_piles = 0
    for i, tag in ((0, "001"), (1, "002")):
        files = sh.overlaps(i)
        ks = [j[0] for j in sh.jobs_of(i)]
        for k, recs in zip(ks, files):   # every job's records == the bytes the reference program wrote for that job
            assert overlap.encode(recs, np.zeros(2, dtype=np.uint32)) == ref_ovl[k], "seed file %d, job %d" % (i, k)
        with open(os.path.join(ra, "in%s.fofn" % tag), "w") as f:
            f.write("\n".join(os.path.join(ra, "%s.%d.ovl" % (os.path.basename(jobs[k][1]), k)) for k in ks) + "\n")
        refpipe.run([R("ovl_sort"), "-m", "2g", "-t", "2", "-k", "17", "-i", os.path.join(db, ".input.seed.%s.idx" % tag), "-o",
                     "ref.%s.sorted.ovl" % tag, "in%s.fofn" % tag], cwd=ra)
        so = os.path.join(ra, "ref.%s.sorted.ovl" % tag)
        bl = {int(l.split()[0]) for l in open(so + ".bl")}
        want = list(refpipe.read_piles(idxs, so, min_len_seed=2000, min_len_aln=500, max_cov_aln=130, min_cov_seed=10, blacklist=bl))
        sub, off, seeds, n_bl = sh.piles(i, files=files)
        assert n_bl == len(bl)
        assert [int(s) for s in seeds] == [w[0] for w in want]
        for p, w in enumerate(want):
            assert np.array_equal(sub[int(off[p]):int(off[p + 1])], w[5]), (i, p)
        n_piles += len(want)
    assert n_piles >= 6


class _PairBackend:
    """A stand-in for the device backend (no GPU, no oracle): one record per (query read, target read) pair whose ids add up to a
    multiple of three, in query order -- enough to see which records a map call returns and in which order.  `concurrent` like the
    device backend: `Shard.overlaps` groups the jobs of one index into one call and runs the other indexes' calls on threads."""
    concurrent = True

    def __init__(self):
        import threading
        self.calls, self.lock, self.released = [], threading.Lock(), []

    def map(self, key, target, query, batch_size, dual):
        from nextdenovo_amd import overlap
        with self.lock:
            self.calls.append((key, len(query), dual))
        out = []
        for q in query.ids.tolist():
            for t in target.ids.tolist():
                if (q + t) % 3 == 0 and (dual or q <= t):
                    out.append((0, q, 1, 2, t, 3, 4, q * 1000 + t))
        return np.asarray(out, dtype=np.uint32).reshape(-1, 8).view(overlap.REC).reshape(-1) if out else np.zeros(0, dtype=overlap.REC)

    def release(self, key, keep_stats=False):
        with self.lock:
            self.released.append(key)


def test_grouped_jobs_with_the_hand_over(tmp_path, monkeypatch):
    """Three ranks of one node in one process (threads), each with the stand-in backend: job by job (NDGPU_STAGE_SERIAL) and
    grouped + side by side give every rank the same records for every job, hand-over included, and the grouped form makes fewer
    calls."""
    import threading
    from nextdenovo_amd import stage
    rng = np.random.default_rng(4)
    lens = rng.integers(600, 9000, 160).astype(np.uint32)
    word_off = np.zeros(lens.size, dtype=np.uint64)
    word_off[1:] = np.cumsum((lens.astype(np.uint64) + 15) // 16)[:-1]
    words = np.zeros(int(((lens.astype(np.uint64) + 15) // 16).sum()) + 1, dtype=np.uint32)
    results = {}
    for mode in ("serial", "grouped"):
        if mode == "serial":
            monkeypatch.setenv("NDGPU_STAGE_SERIAL", "1")
        else:
            monkeypatch.delenv("NDGPU_STAGE_SERIAL")
        xdir = tmp_path / mode
        got, calls, errs = {}, {}, []

        def rank(r):
            try:
                be = _PairBackend()
                sh = stage.Shard(words, word_off, lens, seed_cutoff=3000, read_cutoff=500, n_seed_files=5, backend=be,
                                 exchange=stage.Exchange(str(xdir), r, timeout_s=30.0))
                assert len(sh.part_ids) == 1
                got[r] = [x.copy() for x in sh.overlaps(r)]
                calls[r] = list(be.calls)
            except BaseException as e:   # noqa: BLE001
                errs.append(e)
        ths = [threading.Thread(target=rank, args=(r,)) for r in range(5)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not errs, errs
        results[mode] = (got, calls)
    for r in range(5):
        a, b = results["serial"][0][r], results["grouped"][0][r]
        assert len(a) == len(b) == 6 and sum(x.size for x in a) > 50          # part job + 5 seed x seed jobs per seed file
        for x, y in zip(a, b):
            assert x.tobytes() == y.tobytes()
    n_serial = sum(len(c) for c in results["serial"][1].values())
    n_grouped = sum(len(c) for c in results["grouped"][1].values())
    assert n_serial == 5 + 15 and n_grouped < n_serial                          # every job once across the node; fewer, larger calls


class _SplitBackend(_PairBackend):
    """The stand-in with a target that splits into -I parts as `minimap2_nd.index_parts` cuts it: the whole query set is mapped
    against part 0, then against part 1, ... (`_IndexCache.map`, minimap2/main.c:488-528), so a fused call's records are part-major."""

    def map(self, key, target, query, batch_size, dual):
        from nextdenovo_amd import minimap2_nd, overlap
        with self.lock:
            self.calls.append((key, len(query), dual))
        out = []
        tids = target.ids.tolist()
        for a, b in minimap2_nd.index_parts(target.lens, batch_size):
            for q in query.ids.tolist():
                for t in tids[a:b]:
                    if (q + t) % 3 == 0 and (dual or q <= t):
                        out.append((0, q, 1, 2, t, 3, 4, q * 1000 + t))
        return np.asarray(out, dtype=np.uint32).reshape(-1, 8).view(overlap.REC).reshape(-1) if out else np.zeros(0, dtype=overlap.REC)


def test_grouped_jobs_against_a_target_of_several_index_parts(monkeypatch):
    """when the target seed file splits into more than one -I part and two jobs share the call, the fused
    call's records come back part-major (file 0, 1, 0, 1); every job must still get its own records, part 0 then part 1, exactly
    as its own call would have returned them."""
    from nextdenovo_amd import minimap2_nd, stage
    rng = np.random.default_rng(9)
    lens = rng.integers(600, 9000, 200).astype(np.uint32)
    word_off = np.zeros(lens.size, dtype=np.uint64)
    word_off[1:] = np.cumsum((lens.astype(np.uint64) + 15) // 16)[:-1]
    words = np.zeros(int(((lens.astype(np.uint64) + 15) // 16).sum()) + 1, dtype=np.uint32)
    parts_of = minimap2_nd.index_parts
    monkeypatch.setattr(minimap2_nd, "index_parts", lambda lens_, batch: parts_of(lens_, batch, 20000))
    results = {}
    for mode in ("serial", "grouped"):
        if mode == "serial":
            monkeypatch.setenv("NDGPU_STAGE_SERIAL", "1")
        else:
            monkeypatch.delenv("NDGPU_STAGE_SERIAL", raising=False)
        be = _SplitBackend()
        sh = stage.Shard(words, word_off, lens, seed_cutoff=3000, read_cutoff=500, n_seed_files=5, backend=be)
        sh.seed_batch = 60000           # -I of the seed x seed jobs: every seed file is cut into several parts
        assert len(minimap2_nd.index_parts(lens[sh.seed_ids[1]], sh.seed_batch)) > 1
        results[mode] = ([x.copy() for x in sh.overlaps(1)], list(be.calls))
    a, b = results["serial"][0], results["grouped"][0]
    assert len(a) == len(b) and sum(x.size for x in a) > 50
    for x, y in zip(a, b):
        assert x.tobytes() == y.tobytes()
    assert len(results["grouped"][1]) < len(results["serial"][1])


def test_stage_pipeline_order_depth_and_the_memory_fallback():
    """stage.StagePipeline by itself: results come back in the order the items were given; the piles of item k + 1 are made while
    item k is being corrected; at most `depth` corrections are in flight; a `make_piles` that runs out of device memory beside a
    correction is run again alone when its turn comes, and the line goes on one stage after the other."""
    import threading
    import time
    from nextdenovo_amd import stage
    lock = threading.Lock()
    state = {"correcting": 0, "max_correcting": 0, "made_during_correct": 0, "log": []}

    def make(i):
        with lock:
            if state["correcting"]:
                state["made_during_correct"] += 1
            state["log"].append(("make", i))
        time.sleep(0.02)
        return i * 10

    def correct(i, piles):
        with lock:
            state["correcting"] += 1
            state["max_correcting"] = max(state["max_correcting"], state["correcting"])
        time.sleep(0.05)
        with lock:
            state["correcting"] -= 1
        return piles + 1

    line = stage.StagePipeline(make, correct, depth=2)
    got = list(line.run(range(6)))
    assert [g[0] for g in got] == list(range(6)) and [g[1] for g in got] == [i * 10 + 1 for i in range(6)]
    assert all(a[2] <= b[2] + 1.0 for a, b in zip(got, got[1:]))
    assert state["max_correcting"] == 2 and state["made_during_correct"] >= 3
    assert [x for x in state["log"] if x[0] == "make"] == [("make", i) for i in range(6)]   # every item's piles made once, in order
    # one stage after the other
    state.update(correcting=0, max_correcting=0, made_during_correct=0, log=[])
    got = list(stage.StagePipeline(make, correct, depth=1, prefetch=False).run(range(3)))
    assert [g[1] for g in got] == [1, 11, 21] and state["max_correcting"] == 1 and state["made_during_correct"] == 0
    # out of device memory beside the consensus: made again alone, prefetching off from there on
    state.update(correcting=0, max_correcting=0, made_during_correct=0, log=[])
    failed = []

    def make_oom(i):
        with lock:
            busy = state["correcting"] > 0
        if i == 2 and not failed:   # (the first attempt, made ahead of the line beside whatever is being corrected)
            failed.append(i)
            raise MemoryError("no room beside the consensus")
        with lock:
            state["log"].append(("make", i, busy))
        return i * 10

    line = stage.StagePipeline(make_oom, correct, depth=2)
    got = list(line.run(range(5)))
    assert [g[1] for g in got] == [i * 10 + 1 for i in range(5)] and failed == [2]
    assert line.prefetch is False and line.depth == 1
    assert ("make", 2, False) in state["log"] and ("make", 3, False) in state["log"]   # item 2 again with nothing else running; 3 not prefetched
    # any other failure of make_piles is the caller's
    def make_bad(i):
        if i == 1:
            raise ValueError("a bad option")
        return i
    with pytest.raises(ValueError):
        list(stage.StagePipeline(make_bad, correct, depth=2).run(range(3)))
