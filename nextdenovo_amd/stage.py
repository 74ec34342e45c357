"""The correction stage of one read set in memory, sharded the way nextDenovo shards it.

`seq_dump` deals the seeds (reads >= seed_cutoff) round robin into `seed_cutfiles` seed files and the shorter reads into
part files (util/seq_dump.c:74-114); raw_align runs one `minimap2-nd --step 1` job per (seed file, other file) pair
(nextDenovo:426-467), sort_align one `ovl_sort` and seed_cns one `nextcorrect.py` per seed file (nextDenovo:344-354,
77-84).  A seed file is therefore the unit that shards: everything it needs is the overlaps between its seeds and every
read.  `Shard.piles(i)` computes exactly the jobs whose `.ovl` files the sort of seed file i reads -- its own
(i, part j), (i, seed t >= i), and the mirrors (t < i, seed i) that the reference reaches through `ln -sf`
(nextDenovo:459) -- hands them to `ndgpu_ovl_sort` in job order and applies the pile admission of lib/nextcorrect.py:92-143.
No rank needs anything another rank computed: bench.py --gpus N and the multi-GPU tests give seed file r to rank r.

Computing every mirror on the rank that needs it maps each seed x seed pair twice across the node (N jobs and N index builds per
rank instead of the (N + 1) / 2 jobs the job matrix holds per seed file).  The reference maps a pair once and hands the file over
with `ln -sf` (nextDenovo:455-459); `Shard(exchange=Exchange(dir, rank))` is that hand-over for ranks of one node: every
seed x seed job has ONE owner (`owner_of`: the job (i, t) belongs to rank i when i + t is odd or i == t, to rank t otherwise, so
every rank owns about (N + 1) / 2 jobs), the owner writes the job's records into a node-local directory (/dev/shm), the other
rank that needs them reads them there.  Orientation is the reference's (target = seed file i, query = seed file t) whoever
computes the job, so the records are the same bytes either way; a file that does not arrive in time is computed locally.
"""
from __future__ import annotations

import os
import time

import numpy as np

from . import minimap2_nd, overlap
from .correct_stage import _IndexCache, job_matrix


class DeviceBackend:
    """The product path: indexes and maps on the device (libndgpu_overlap.so), sorts with ndgpu_ovl_sort."""
    concurrent = True   # map calls against different indexes may run side by side (an index has its own stream)

    def __init__(self, opt):
        self.opt, self.caches, self.last_stats, self.calls = opt, {}, None, 0
        self._lock = __import__("threading").Lock()   # (the cache dictionary; an index itself is used by one thread at a time)

    def map(self, key, target, query, batch_size, dual):
        with self._lock:
            if key not in self.caches:
                self.caches[key] = _IndexCache(self.opt, target)
            c = self.caches[key]
            self.calls += 1
        return c.map(query, batch_size, dual)

    def release(self, key, keep_stats=False):
        with self._lock:
            c = self.caches.pop(key, None)
        if c is not None:
            if keep_stats:
                st = [ix.stats() for ix in c.parts.values()]
                self.last_stats = {k_: sum(s[k_] for s in st) for k_ in st[0]} if st else None
            c.close()

    def sort(self, files, seed_len, min_seed_len, k, flank):
        return overlap.sort_overlaps(files, seed_len, min_seed_len, k, flank)


def deal(lens: np.ndarray, read_cutoff: int, seed_cutoff: int, n_seed_files: int, block_size: int = 0):
    """Read ids (dense, in input order over the reads >= read_cutoff) of every seed file and part file, as seq_dump
    deals them (util/seq_dump.c:74-114; the FILE_MAX_SIZE overflow files are not modelled: 2^31-ish bases per file)."""
    lens = np.asarray(lens, dtype=np.int64)
    kept = np.nonzero(lens >= read_cutoff)[0]
    ids = np.arange(kept.size, dtype=np.uint32)
    is_seed = lens[kept] >= seed_cutoff
    seed_ids = ids[is_seed]
    seed_files = [seed_ids[k::n_seed_files] for k in range(n_seed_files)]
    part_ids = ids[~is_seed]
    parts = []
    if part_ids.size:
        if block_size <= 0:
            parts = [part_ids]
        else:  # a part file closes once its total would exceed the block size
            tot, lo = 0, 0
            pl = lens[kept][~is_seed]
            for j in range(part_ids.size):
                tot += int(pl[j])
                if tot > block_size:
                    parts.append(part_ids[lo:j])
                    lo, tot = j, int(pl[j])
            parts.append(part_ids[lo:])
    return kept, seed_files, parts


def owner_of(i: int, t: int) -> int:
    """The rank that computes the seed x seed job (target seed file i, query seed file t), i <= t."""
    return i if (i == t or (i + t) % 2 == 1) else t


def plan(n_seed_files: int, n_part_files: int = 0):
    """Per rank (= seed file): the raw_align jobs it computes and the index builds they need, without and with the hand-over --
    what bounds the overlap stage's share of a multi-GPU step.  Jobs cost about the same (1 / N^2 of the pair space each)."""
    rows = []
    for r in range(n_seed_files):
        needs = [j for j in job_matrix(n_seed_files, n_part_files) if j[1] == r or (j[2] == "seed" and j[3] == r)]
        own = [j for j in needs if j[2] == "part" or owner_of(j[1], j[3]) == r]
        rows.append({"rank": r, "jobs_needed": len(needs), "jobs_computed_alone": len(needs), "index_builds_alone": len({j[1] for j in needs}),
                     "jobs_computed_with_exchange": len(own), "index_builds_with_exchange": len({j[1] for j in own}),
                     "jobs_received": len(needs) - len(own)})
    return rows


class Exchange:
    """Node-local hand-over of seed x seed job records between the ranks of one node (one directory, one file per job and step)."""

    def __init__(self, directory: str, rank: int, timeout_s: float = 120.0, lookahead: int = 0):
        """lookahead: how many steps beyond the one a reader waits for its owner may have written files of WITHOUT being through with that
        step -- 0 for an owner that makes its steps one after the other; `StagePipeline.ahead` for an owner whose steps are made by
        several producers side by side (each with an Exchange object of its own on the same directory, steps numbered by the caller)."""
        self.dir, self.rank, self.timeout_s, self.step = directory, rank, timeout_s, 0
        self.lookahead = max(0, int(lookahead))
        self.KEEP_STEPS = 3 + 2 * self.lookahead
        os.makedirs(directory, exist_ok=True)
        self.stats = {"sent": 0, "received": 0, "recomputed": 0, "wait_s": 0.0}
        self._mine = []   # (step, path) of the files this rank wrote and nobody is known to have read

    KEEP_STEPS = 3   # an owner keeps the files of its last steps; older ones nobody has read are removed by it

    def _path(self, step, k, owner):
        return os.path.join(self.dir, "step%06d.job%05d.r%d.npy" % (step, k, owner))

    def begin_step(self, step=None):
        """One call per LOGICAL step (`Shard.piles` makes it once, before its attempts: a retry after an out-of-memory overlap
        stage stays in the step its peers are in).  A file is removed by its one reader once it has been read -- the writer may be
        steps ahead of the reader --, and what a reader never came for is removed by its owner KEEP_STEPS steps later."""
        self.step = self.step + 1 if step is None else int(step)
        keep = []
        for st, path in self._mine:
            if st <= self.step - self.KEEP_STEPS:
                try:
                    os.unlink(path)
                except OSError:
                    pass
            else:
                keep.append((st, path))
        self._mine = keep

    def put(self, k, recs, owner=None):
        """Job k of this step, written under the SEED FILE that owns it (`owner_of`; the rank by default: bench.py gives seed file r to
        rank r)."""
        final = self._path(self.step, k, self.rank if owner is None else owner)
        tmp = final + ".tmp"
        with open(tmp, "wb") as f:
            np.save(f, recs)
        os.rename(tmp, final)   # (atomic: a reader sees the whole file or none)
        self._mine.append((self.step, final))
        self.stats["sent"] += 1

    def _owner_moved_on(self, k, owner):
        """The owner has written job k of a LATER step: the file of this step is not coming any more (it was read by nobody in time
        and removed, or the owner never wrote it)."""
        import glob
        for path in glob.glob(os.path.join(self.dir, "step*.job%05d.r%d.npy" % (k, owner))):
            try:
                if int(os.path.basename(path)[4:10]) > self.step + self.lookahead:
                    return True
            except ValueError:
                pass
        return False

    def get(self, k, owner):
        """The records of job k as the owner (a seed file index, `owner_of`) wrote them, or None after the timeout / once the owner
        is seen to have moved on."""
        path = self._path(self.step, k, owner)
        t0 = time.perf_counter()
        t_look = t0
        while not os.path.exists(path):
            now = time.perf_counter()
            if now - t0 > self.timeout_s or (now - t_look > 0.05 and self._owner_moved_on(k, owner) and not os.path.exists(path)):
                self.stats["wait_s"] += now - t0
                return None
            if now - t_look > 0.05:
                t_look = now
            time.sleep(0.0005)
        self.stats["wait_s"] += time.perf_counter() - t0
        self.stats["received"] += 1
        recs = np.load(path)
        try:
            os.unlink(path)   # (a seed x seed job has exactly one other reader)
        except OSError:
            pass
        return recs


class StagePipeline:
    """Seed files through the stage like parts through a line: while the consensus of seed file k holds the device, the piles of seed
    file k + 1 -- overlap jobs, sort, pile admission, whatever `make_piles` does -- are computed on a thread of their own, and up to
    `depth` consensus calls are in flight: a call ends with its last sub-batches' low-quality-region stages (host ranking and POA, two
    rounds of small launches) and little else on the device, which the next call's main phases fill.  The library's contexts serve the
    older call first (DeviceAligner::begin_batch(order)).  The reference runs the same stage seed file by seed file, several at a time
    (nextDenovo:344-354 with `pa_correction` subtasks side by side); one GPU takes them one behind the other.

    make_piles(item) -> piles;  correct(item, piles) -> result.  `run` yields (item, result, t_end) in the order the items were given.
    A `make_piles` that runs out of device memory beside the consensus (MemoryError) is run again when its turn comes, alone, and
    prefetching stays off from there on (genome-scale read sets whose two stages do not fit the device side by side)."""

    def __init__(self, make_piles, correct, depth=2, prefetch=True, producers=1, ahead=2):
        self.make_piles, self.correct = make_piles, correct
        self.depth = max(1, int(depth)) if prefetch else 1
        self.prefetch = bool(prefetch)
        self.producers = max(1, int(producers))   # threads that make piles side by side (make_piles(item) must then be safe to call from two threads
        self.ahead = max(1, int(ahead))           # for DIFFERENT items: e.g. a Shard per thread); sets of piles made ahead of the line at most
        self.wait_s = 0.0            # time the line stood still waiting for prefetched piles

    def run(self, items):
        import sys
        import threading
        from concurrent.futures import ThreadPoolExecutor
        items = list(items)

        def timed(it, piles):
            r = self.correct(it, piles)
            return r, time.perf_counter()

        # the producers run ahead of the line on their own: the piles of a later item are begun when a producer is free (not when the line
        # comes to take them), at most `ahead` sets beyond the item the line is at -- so the line never stands still for a stage that could
        # have begun earlier.  (Calls that share the device end in pairs as often as one by one, and the line then asks for two sets in
        # quick succession.)
        cond = threading.Condition()
        state = {"taken": 0, "next": 1, "made": {}, "stop": False}

        def produce():
            while True:
                with cond:
                    while not state["stop"] and (state["next"] >= len(items) or state["next"] > state["taken"] + self.ahead):
                        if state["next"] >= len(items):
                            return
                        cond.wait(0.05)
                    if state["stop"]:
                        return
                    n = state["next"]
                    state["next"] += 1
                try:
                    res = ("ok", self.make_piles(items[n]))
                except BaseException as e:   # noqa: BLE001  (taken up by the turn that wants the piles)
                    res = ("error", e)
                with cond:
                    state["made"][n] = res
                    if res[0] == "error":
                        state["stop"] = True
                    cond.notify_all()
                if res[0] == "error":
                    return

        threads = []
        inflight = []
        try:
            with ThreadPoolExecutor(max_workers=self.depth) as pool:
                for n, it in enumerate(items):
                    if threads and n >= 1:
                        t0 = time.perf_counter()
                        with cond:
                            while n not in state["made"]:
                                if state["stop"] and n >= state["next"]:   # (a producer failed on an earlier item and nobody will make this one)
                                    break
                                cond.wait(0.05)
                            kind, val = state["made"].pop(n, ("missing", None))
                            state["taken"] = n
                            cond.notify_all()
                        self.wait_s += time.perf_counter() - t0
                        if kind == "error" and isinstance(val, MemoryError):
                            sys.stderr.write("[ndgpu stage] the next seed file's overlap stage ran out of device memory beside the consensus: "
                                             "one stage after the other from here on\n")
                            self.prefetch = False
                            self.depth = 1
                            for th in threads:
                                th.join()
                            threads = []
                            with cond:
                                state["made"].clear()      # (sets made ahead are dropped: each is made again at its turn, alone)
                            while inflight:            # (its turn has come: alone on the device)
                                i0, f0 = inflight.pop(0)
                                r0, te0 = f0.result()
                                yield i0, r0, te0
                            piles = self.make_piles(it)
                        elif kind == "error":
                            raise val
                        elif kind == "missing":
                            piles = self.make_piles(it)
                        else:
                            piles = val
                    else:
                        piles = self.make_piles(it)
                        if self.prefetch and len(items) > 1 and n == 0:   # (the first item's piles are the line's own: nothing runs before the line does)
                            threads = [threading.Thread(target=produce) for _ in range(self.producers)]
                            for th in threads:
                                th.start()
                    inflight.append((it, pool.submit(timed, it, piles)))
                    while len(inflight) >= self.depth:
                        i0, f0 = inflight.pop(0)
                        r0, te0 = f0.result()
                        yield i0, r0, te0
                while inflight:
                    i0, f0 = inflight.pop(0)
                    r0, te0 = f0.result()
                    yield i0, r0, te0
        finally:
            with cond:
                state["stop"] = True
                cond.notify_all()
            for th in threads:
                th.join()


class Shard:
    """One read set (2-bit words as in a .2bit file, read i at words[word_off[i]], lens[i] bases; read id = index) and
    the parameters of the stage."""

    def __init__(self, words, word_off, lens, preset="ava-ont", seed_cutoff=1000, read_cutoff=500, n_seed_files=1, sort_k=40,
                 flank=300, min_len_seed=None, min_len_aln=500, max_cov_aln=130, min_cov_seed=10, blacklist=True, occ=None,
                 backend=None, exchange=None):
        self.words = np.ascontiguousarray(words, dtype=np.uint32)
        self.word_off = np.ascontiguousarray(word_off, dtype=np.uint64)
        self.lens = np.ascontiguousarray(lens, dtype=np.uint32)
        kept, self.seed_ids, self.part_ids = deal(self.lens, read_cutoff, seed_cutoff, n_seed_files)
        if kept.size != self.lens.size:
            raise ValueError("reads below read_cutoff must be dropped by the caller (ids are dense over the kept reads)")
        argv = ["--step", "1", "-x", preset] + (["-f", str(occ)] if occ else []) + ["a", "b"]
        self.opt = minimap2_nd.build_opt(minimap2_nd.parse_argv(argv))
        self.seed_batch = 6000000000 if preset == "ava-hifi" else 3000000000   # -I of the seed x seed jobs (nextDenovo:430,456)
        self.sort_k, self.flank = sort_k, flank
        self.min_len_seed = seed_cutoff // 2 if min_len_seed is None else min_len_seed   # nextDenovo passes seed_cutoff / 2... see config_parser
        self.min_len_aln, self.max_cov_aln, self.min_cov_seed, self.blacklist = min_len_aln, max_cov_aln, min_cov_seed, blacklist
        self.stats = {"overlap_s": 0.0, "sort_s": 0.0, "assemble_s": 0.0, "records": 0, "jobs": 0}
        self.ovl_stats = None
        # tests plug a CPU backend built from oracle/ here (same two calls); the product has the device backend only
        self.backend = backend if backend is not None else DeviceBackend(self.opt)
        self.exchange = exchange   # None: every job this seed file needs is computed here
        # the reads are mapped step after step and job after job: their words go to the device once (like the consensus read DB),
        # not with every index build and every query batch
        # (the device copy is keyed by this array's address range: the array must not be written to while the Shard lives -- it is
        # made read-only here where that is this object's to decide)
        self._resident = False
        if isinstance(self.backend, DeviceBackend) and not os.environ.get("NDGPU_OVL_NO_RESIDENT"):   # (the switch: an A/B knob)
            try:
                overlap.words_resident(self.words)
                self._resident = True
                if self.words.flags.owndata:
                    self.words.flags.writeable = False
            except (MemoryError, RuntimeError) as e:   # no room for the copy: every index build / query batch uploads its words
                import sys
                print("[ndgpu stage] read words not resident (%s): uploaded per call" % (e,), file=sys.stderr)

    def close(self):
        if self._resident:
            overlap.words_release(self.words)
            self._resident = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _set(self, ids):
        return overlap.ReadSet(ids, self.lens[ids], self.words, self.word_off[ids])

    def jobs_of(self, i):
        """[(k, target seed file, query kind, query index, dual)] whose records the sort of seed file i reads, in job order."""
        return [j for j in job_matrix(len(self.seed_ids), len(self.part_ids))
                if j[1] == i or (j[2] == "seed" and j[3] == i)]

    def overlaps(self, i, own_step=False):
        """Step-1 records of every job of seed file i, job order.  own_step: the caller has begun the exchange's step itself (a retry
        must not begin another)."""
        import time
        t0 = time.perf_counter()
        out = []
        be = self.backend
        ex = self.exchange
        if ex is not None and not own_step:
            ex.begin_step()
        try:
            jobs = self.jobs_of(i)

            def compute(k, tgt, kind, j, dual):
                query = self._set(self.part_ids[j] if kind == "part" else self.seed_ids[j])
                r = be.map(tgt, self._set(self.seed_ids[tgt]), query, minimap2_nd.IDX_BATCH if kind == "part" else self.seed_batch, dual)
                if tgt != i:  # that index is not needed again by this shard
                    be.release(tgt)
                return r
            got = {}
            # the jobs this rank owns first (the others wait for them), then what the others hand over
            own = [job for job in jobs if ex is None or job[2] == "part" or owner_of(job[1], job[3]) == i]

            def hand_over(k, tgt, kind, j):
                if ex is not None and kind == "seed" and tgt != j:
                    ex.put(k, got[k], i)
            if getattr(be, "concurrent", False) and len(own) > 2 and not os.environ.get("NDGPU_STAGE_SERIAL"):
                self._overlaps_grouped(i, own, got, hand_over)
            else:
                for k, tgt, kind, j, dual in own:
                    got[k] = compute(k, tgt, kind, j, dual)
                    hand_over(k, tgt, kind, j)
            for k, tgt, kind, j, dual in jobs:
                if k not in got:
                    r = ex.get(k, owner_of(tgt, j))
                    if r is None:   # the owner is late or gone: the job is computed here, the step goes on
                        ex.stats["recomputed"] += 1
                        r = compute(k, tgt, kind, j, dual)
                    got[k] = r
            out = [got[k] for k, *_ in jobs]
        finally:
            be.release(i, keep_stats=True)
            self.ovl_stats = getattr(be, "last_stats", None)
            for key in list(getattr(be, "caches", {})):   # a failed attempt must not leave the indexes of its mirror jobs cached
                be.release(key)
        self.stats["overlap_s"] += time.perf_counter() - t0
        self.stats["records"] += int(sum(r.size for r in out))
        self.stats["jobs"] += len(out)
        return out

    def _overlaps_grouped(self, i, own, got, hand_over):
        """The jobs a rank of many computes are small (1 / N^2 of the pair space each) and a map call has a floor -- the chain and the
        replayed sort of its heaviest read, a dozen waits -- that does not shrink with the job: at N = 8 the five to nine calls of a
        rank took 62 ms where the whole job's ONE call takes 86.  So (a) the jobs that map against the same index with the same options
        go into ONE call -- their query files back to back; the reads of a query set are mapped independently of each other, so the
        call's records are the jobs' records back to back, cut apart again at the file boundaries -- and (b) the calls against
        DIFFERENT indexes (the mirror jobs: target = another rank's seed file) run side by side, a host thread and an index -- with
        its stream -- each.  An index is never used by two threads at a time."""
        import threading
        be = self.backend
        by_target = {}
        for job in own:
            k, tgt, kind, j, dual = job
            batch = minimap2_nd.IDX_BATCH if kind == "part" else self.seed_batch
            layout = tuple(minimap2_nd.index_parts(self.lens[self.seed_ids[tgt]], batch))   # (jobs of one call share the -I split)
            by_target.setdefault(tgt, {}).setdefault((dual, layout), (batch, []))[1].append(job)   # (-I only matters through the split)
        errors = []
        hand_lock = threading.Lock()   # (the hand-over keeps a list and counters: one thread at a time)

        def run_target(tgt, groups):
            try:
                target = self._set(self.seed_ids[tgt])
                for (dual, _layout), (batch, members) in groups.items():
                    ids = [self.part_ids[j] if kind == "part" else self.seed_ids[j] for _k, _t, kind, j, _d in members]
                    if len(members) == 1:
                        parts = [be.map(tgt, target, self._set(ids[0]), batch, dual)]
                    else:
                        recs = be.map(tgt, target, self._set(np.concatenate(ids)), batch, dual)
                        # a record's query is its qname (lib/ovl.h:20-25): which file of the call it came from
                        file_of = np.zeros(self.lens.size, dtype=np.int32)
                        for f, x in enumerate(ids):
                            file_of[x] = f
                        rf = file_of[recs["qname"]] if recs.size else np.zeros(0, dtype=np.int32)
                        # a target that splits into several -I parts is mapped part by part, the whole query set each time
                        # (minimap2/main.c:488-528): the call's records are part-major -- file 0..n of part 0, file 0..n of part
                        # 1, ... -- and a job's own file is its records of part 0, part 1, ... in that order.  A stable sort by
                        # file gives exactly that (the identity when the target is one part).
                        if recs.size and np.any(np.diff(rf) < 0):
                            order = np.argsort(rf, kind="stable")
                            recs, rf = recs[order], rf[order]
                        cut = np.searchsorted(rf, np.arange(len(ids) + 1))
                        parts = [recs[cut[f]:cut[f + 1]] for f in range(len(ids))]
                    for (k, _t, kind, j, _d), r in zip(members, parts):
                        got[k] = r
                        with hand_lock:
                            hand_over(k, tgt, kind, j)
                if tgt != i:   # that index is not needed again by this shard
                    be.release(tgt)
            except BaseException as e:   # noqa: BLE001  (re-raised by the caller's thread)
                errors.append(e)
        threads = [threading.Thread(target=run_target, args=(tgt, groups)) for tgt, groups in by_target.items() if tgt != i]
        for th in threads:
            th.start()
        if i in by_target:
            run_target(i, by_target[i])
        for th in threads:
            th.join()
        if errors:
            raise errors[0]

    def piles(self, i, files=None, step=None):
        """(records [n, 8] uint32, pile_off, seed ids, blacklisted) of seed file i: what `nextcorrect.py -i sorted.ovl` corrects.
        step: the hand-over's number of this logical step (default: one more than this Shard's last) -- given by a caller that has the
        steps of one rank made by several Shards side by side."""
        import time
        if files is None:
            from . import api
            if getattr(self, "_release_first", False):
                api.release_device_memory()   # learned below: on this device the two stages do not fit side by side
            if self.exchange is not None:
                self.exchange.begin_step(step)   # once per logical step, whatever happens below
            try:
                files = self.overlaps(i, own_step=True)
            except MemoryError:
                # the overlap stage ran out of device memory (any other failure -- a bad option, a failed occurrence-threshold
                # query -- is raised as it is): the consensus contexts still hold the buffers of the last call
                # (they keep them between calls on purpose).  Hand those back, run the stage once more, and from now on hand
                # them back before the stage starts instead of finding out halfway through it.
                if isinstance(self.backend, DeviceBackend) and api.release_device_memory() > 0:
                    overlap.trim()
                    self._release_first = True
                    files = self.overlaps(i, own_step=True)
                else:
                    raise
        t0 = time.perf_counter()
        seed_len = np.zeros(self.lens.size, dtype=np.uint32)
        sid = self.seed_ids[i]
        seed_len[sid] = self.lens[sid]
        srt, bl, sst = self.backend.sort(files, seed_len, int(self.lens[sid].min()) if sid.size else 0, self.sort_k, self.flank)
        t1 = time.perf_counter()
        skip = [rid for rid, _ in bl] if self.blacklist else []
        sub, off, seeds = overlap.assemble_piles(srt, int(self.lens.size), self.min_len_seed, self.min_len_aln, self.max_cov_aln,
                                                 self.min_cov_seed, skip)
        self.stats["sort_s"] += t1 - t0
        self.stats["assemble_s"] += time.perf_counter() - t1
        self.sort_stats = sst
        if isinstance(self.backend, DeviceBackend):
            # the consensus contexts size their (grow-only) buffers from what is free when they are called: leave the overlap stage
            # of the next seed file what this one needed beyond what the library still has cached
            from . import api
            # (+ a margin for what the overlap stage takes outside its block pool: scratch of spilling kernels, rocPRIM, the runtime)
            _live, cached, peak = overlap.pool_bytes()
            # (the margin scales with the device: 32 GB on a 288 GB MI355X, an eighth of the memory on a smaller part -- a fixed 32 GB
            # left the consensus stage its 4 GB floor on anything below ~50 GB)
            total = getattr(self, "_device_total", 0)
            if not total:   # (asked once: the query goes through the runtime's device enumeration, 3 ms of every step)
                try:
                    import torch
                    total = torch.cuda.mem_get_info()[1] if torch.cuda.is_available() else 288 << 30
                except Exception:
                    total = 288 << 30
                self._device_total = total
            api.reserve_device_memory(max(0, peak - cached) + max(min(32 << 30, total // 8), peak // 4))
        return sub, off, seeds, len(bl)
