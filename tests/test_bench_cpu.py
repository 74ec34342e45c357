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
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["cpu_baseline"]["kind"] == "reference" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert d["parity"]["piles"] > 5 and d["parity"]["mismatch"] == 0
    assert d["overlap"]["cpu_baseline"]["device_ovl_identical"] is True
    # the timed step followed a warm-up step: nothing was (re)allocated in it
    assert d["allocations"]["in_step"] == 0 and d["overlap"]["pool_calls"]["n"] == 0
    assert d["counters"]["piles"] == d["config"]["piles_rank0"] and d["counters"]["lq_declined"] == 0
