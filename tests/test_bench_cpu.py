"""`bench.py` itself without a GPU: its main() on a 30 kb genome with the two libraries bound to the interpreted builds (tests/simt) --
the overlap -> sort -> pile assembly -> consensus step, the JSON contract of the one line it prints (metric, roofline, cpu_baseline,
parity, allocation counters), the parity block against the compiled reference and the overlap job's byte comparison."""
import ctypes as C
import io
import json
import os
import sys
from contextlib import redirect_stdout

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "simt"))
sys.path.insert(0, ROOT)


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "nextcorrect.so")), reason="oracle/_ref not built")
def test_bench_line_on_a_tiny_genome(monkeypatch):
    import build_simt
    from nextdenovo_amd import api, overlap
    monkeypatch.setenv("NDGPU_CONTEXTS", "2")
    monkeypatch.setenv("NDGPU_OVL_SLAB_MB", "64")   # the overlap library's block pool as on the device: slabs, carved up by the library
    monkeypatch.setattr(overlap, "_lib", overlap._bind(C.CDLL(build_simt.build_overlap())))
    monkeypatch.setattr(api, "_LIB", api._bind(C.CDLL(build_simt.build())))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--genome-size", "30000", "--depth", "14", "--steps", "1", "--warmup", "1"])
    import bench
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                    # ONE JSON line
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "corrected bases/sec" and d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "synthetic uniform 0.03 Mb" in d["config"]["workload"] and "kernel" not in d["config"]["workload"]   # the read set's name, nothing else
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["cpu_baseline"]["kind"] == "reference" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    cb = d["cpu_baseline"]   # both rates: the pool's makespan and the per-core figure measured inside the workers
    assert cb["cpu_seconds"] > 0 and abs(cb["per_core_measured_x_cores"] - cb["per_core_measured"] * cb["cores"]) < 1e-6 * cb["per_core_measured_x_cores"]
    assert cb["wall_s"] > 0 and cb["slowest_pile_cpu_s"] <= cb["cpu_seconds"]
    assert d["parity"]["piles"] > 5 and d["parity"]["mismatch"] == 0
    assert d["parity"]["fasta_records"] > 5 and "cns.fasta" in d["parity"]["compared"]   # the bytes compared are the ones the step wrote
    assert d["overlap"]["cpu_baseline"]["device_ovl_identical"] is True
    # SURVEY 8(d)'s CPU protocol: three runs and their median; the reference's ovl_sort timed on the reference overlapper's file, the
    # device's sort of the device's records byte for byte the same; the metric over raw_align + sort_align + seed_cns
    assert len(cb["runs"]) == 3 and sorted(r["wall_s"] for r in cb["runs"])[1] == round(cb["wall_s"], 3)
    osrt = d["overlap"]["cpu_baseline"]["ovl_sort"]
    assert osrt["wall_s"] > 0 and osrt["sorted_ovl_bytes"] > 100 and osrt["device_sorted_ovl_identical"] is True
    assert cb["stage_chain"]["corrected_bases_per_s"] > 0 and cb["stage_chain"]["corrected_bases_per_s"] < cb["value"] * 1.0001
    # every timed step by itself, the box the host phases ran on, and the stage's output inside the timed region
    assert len(d["step_ms"]["list"]) == d["steps"] and d["step_ms"]["min"] <= d["step_ms"]["median"] <= d["step_ms"]["max"]
    assert d["host"]["cpu_count"] >= 1 and d["host"]["host_threads"] >= 1
    assert d["fasta_write"]["included_in_value"] is True and d["fasta_write"]["bytes_per_step"] > 0
    assert "cns.fasta" in r["timed_step_note"]
    # the timed step followed a warm-up step: nothing was (re)allocated in it
    assert d["allocations"]["in_step"] == 0 and d["overlap"]["pool_calls"]["n"] == 0
    assert d["counters"]["piles"] == d["config"]["piles_rank0"] and d["counters"]["lq_declined"] == 0


def test_bench_steps_in_a_pipeline(monkeypatch):
    """Two timed steps: the overlap / sort / admission stage of step k + 1 runs on a thread of its own during the consensus of step
    k, and two consensus calls are in flight at a time (the contexts of the library serve both) -- the same records, the same parity
    block read back from the file of the step that ended last, K overlap stages and K consensus stages inside the timed region; and
    the same bases per step as with one stage after the other."""
    import build_simt
    from nextdenovo_amd import api, overlap
    monkeypatch.setenv("NDGPU_CONTEXTS", "2")
    monkeypatch.setenv("NDGPU_OVL_SLAB_MB", "64")
    monkeypatch.setattr(overlap, "_lib", overlap._bind(C.CDLL(build_simt.build_overlap())))
    monkeypatch.setattr(api, "_LIB", api._bind(C.CDLL(build_simt.build())))
    import bench
    got = {}
    for mode, extra in (("pipeline", []), ("serial", ["--no-pipeline"])):
        monkeypatch.setattr(sys, "argv", ["bench.py", "--genome-size", "30000", "--depth", "14", "--steps", "2", "--warmup", "1",
                                          "--no-cpu-baseline"] + extra)
        buf = io.StringIO()
        with redirect_stdout(buf):
            bench.main()
        lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        got[mode] = json.loads(lines[0])
    a, b = got["pipeline"], got["serial"]
    assert a["pipeline"]["next_stage_begins_during_consensus"] is True and a["pipeline"]["consensus_calls_in_flight"] == 2
    assert b["pipeline"]["next_stage_begins_during_consensus"] is False and b["pipeline"]["consensus_calls_in_flight"] == 1
    for d in (a, b):
        assert len(d["step_ms"]["list"]) == 2 and d["counters"]["piles"] == 2 * d["config"]["piles_rank0"]
    # the same corrected bases per step either way (value x wall = bases)
    assert abs(a["value"] * a["ms_per_step"] - b["value"] * b["ms_per_step"]) < 1e-6 * a["value"] * a["ms_per_step"]
    assert a["fasta_write"]["bytes_per_step"] == b["fasta_write"]["bytes_per_step"] > 0


def _rank_main(rank, port, q, extra=()):
    """bench.main() of one rank of two, libraries = the interpreted builds, collectives over gloo."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      NDGPU_BENCH_DIST_BACKEND="gloo", NDGPU_DEVICE="0", NDGPU_CONTEXTS="1")
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, "simt"))
    sys.path.insert(0, ROOT)
    import build_simt
    from nextdenovo_amd import api, overlap
    overlap._lib = overlap._bind(C.CDLL(build_simt.build_overlap()))
    api._LIB = api._bind(C.CDLL(build_simt.build()))
    sys.argv = ["bench.py", "--gpus", "2", "--genome-size", "30000", "--depth", "14", "--steps", "2", "--warmup", "1"] + list(extra)   # (two steps: the second one's piles are prefetched, hand-over included, and two consensus calls are in flight)
    import bench
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    q.put((rank, [ln for ln in buf.getvalue().splitlines() if ln.startswith("{")]))


@pytest.mark.parametrize("producers", [1, 2])
def test_bench_line_of_two_ranks(tmp_path, producers):
    """(producers = 2: every rank's later steps are made by two Shards side by side, each with an Exchange object of its own on the
    node's directory and the steps numbered by bench.py -- three timed steps, so that both are at work.)
    `bench.py --gpus 2` as the driver launches it (one process per rank, RANK / WORLD_SIZE / MASTER_* in the environment), on
    the CPU: rank 0 prints the ONE line, whole-job value, strong scaling, per-rank stage times; the seed x seed pair of the two seed
    files is mapped by its owner only and handed over."""
    import socket

    import torch.multiprocessing as mp
    import build_simt
    build_simt.build(), build_simt.build_overlap()   # (not twice at the same time in the children)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    # what a run that died on this port would have left behind: job files of another layout (bench.py starts from an empty directory)
    xdir = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", "ndgpu_bench_%d" % port)
    os.makedirs(xdir, exist_ok=True)
    for k in range(6):
        with open(os.path.join(xdir, "step000001.job%05d.r0.npy" % k), "wb") as f:
            f.write(b"not the records of this job")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    extra = ("--producers", "2", "--steps", "3") if producers == 2 else ()   # (a later --steps wins)
    procs = [ctx.Process(target=_rank_main, args=(r, port, q, extra)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=1200) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[1] == [] and len(res[0]) == 1
    d = json.loads(res[0][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["config"]["seed_files"] == 2
    assert len(d["per_rank"]) == 2 and all(r["piles"] > 0 for r in d["per_rank"])
    ra = d["config"]["raw_align_jobs"]
    n_steps = 3 if producers == 2 else 2
    assert d["steps"] == n_steps and ra["rank0_piles_producers"] == producers
    assert ra["exchange"] is True and ra["per_rank_jobs_computed"] == [2, 1] and ra["rank0_exchange"]["sent"] == 1 + n_steps   # (warm-up + the steps)
    assert ra["rank0_exchange"]["recomputed"] == 0
    assert not os.path.exists(xdir)   # (removed by rank 0 once every rank is past its last read)
