#!/bin/bash
# GPU-box sessions, in stages; every stage leaves its evidence under gpurun_out/<tag>/ even if a later one is cut off.
#   gpurun --timeout 1800 -- 'bash tools/gpu_session.sh <tag> <stages...>'
# stages: tests (full pytest -m gpu, no -x) | testsx (-x) | chain | pressure | bench | bench20 | prof | pmc | modes | c3 | c3shard | rccl | cand
set -u
tag=${1:-r4}; shift || true
stages=${*:-tests bench}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
for s in $stages; do
  echo "== stage $s =="
  t0=$(date +%s)
  case $s in
    tests)  timeout 1700 python -m pytest tests -m gpu -q --durations=20 --timeout=900 > "$out/pytest.log" 2>&1; echo "pytest exit $?" | tee -a "$out/pytest.log"; tail -15 "$out/pytest.log" ;;
    testsx) timeout 1700 python -m pytest tests -m gpu -q -x --timeout=900 > "$out/pytest_x.log" 2>&1; echo "pytest -x exit $?" | tee -a "$out/pytest_x.log"; tail -5 "$out/pytest_x.log" ;;
    chain)  timeout 300 python tools/chain_latency.py "$out/chain_latency.json" > "$out/chain.log" 2>&1; tail -4 "$out/chain.log" ;;
    pressure) timeout 600 python tools/pressure_overlap.py 4.6e6 "$out/pressure_overlap.json" > "$out/pressure.log" 2>&1; echo "pressure exit $?"; grep -v "^\[ndgpu_overlap\]" "$out/pressure.log" | tail -20 ;;
    bench)  timeout 900 python bench.py --steps 10 --warmup 3 > "$out/bench_config2.json" 2> "$out/bench_config2.err"; echo "bench exit $?"
            python - "$out/bench_config2.json" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(" ms_per_step %.1f value %.1f M cns %.1f ovl %.1f | kernel_ms %s | parity %s" % (d["ms_per_step"], d["value"] / 1e6, d["consensus_ms_per_step"], d["overlap"]["ms_per_step"], d["kernel_ms"], d.get("parity", {}).get("mismatch")))
print(" roofline", json.dumps(d["roofline"])[:1500])
P
            ;;
    bench20) timeout 900 python bench.py --steps 20 --warmup 5 > "$out/bench_config2_20.json" 2> "$out/bench_config2_20.err"; echo "bench20 exit $?"; tail -c 600 "$out/bench_config2_20.json" ;;
    quick)  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$out/bench_quick.json" 2> "$out/bench_quick.err"; echo "quick exit $?"
            python - "$out/bench_quick.json" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(" ms_per_step %.1f value %.1f M cns %.1f ovl %.1f | kernel_ms %s" % (d["ms_per_step"], d["value"] / 1e6, d["consensus_ms_per_step"], d["overlap"]["ms_per_step"], d["kernel_ms"]))
P
            ;;
    q:*)    # quick bench under an environment: stage name "q:TAG:VAR=val,VAR2=val2"
            qtag=$(echo "$s" | cut -d: -f2); qenv=$(echo "$s" | cut -d: -f3- | tr ',' ' ')
            env $qenv timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > "$out/bench_q_$qtag.json" 2> "$out/bench_q_$qtag.err"; echo "q $qtag exit $?"
            python tools/bench_brief.py "$out/bench_q_$qtag.json" ;;
    prof)   (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/stats" -o s -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OLDPWD/$out/prof.log" 2>&1)
            python tools/rocprof_summary.py "$(ls "$out"/stats/*results.db | head -1)" > "$out/kernel_stats.txt" 2>> "$out/prof.log"; head -30 "$out/kernel_stats.txt"; rm -rf "$out/stats" ;;
    pmc)    for c in FETCH_SIZE WRITE_SIZE; do
              (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c -d "$OLDPWD/$out/pmc_$c" -o p -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OLDPWD/$out/pmc_$c.log" 2>&1)
            done
            python tools/rocprof_pmc_summary.py "$(ls "$out"/pmc_FETCH_SIZE/*results.db | head -1)" "$(ls "$out"/pmc_WRITE_SIZE/*results.db | head -1)" > "$out/pmc_summary.txt" 2>&1; head -40 "$out/pmc_summary.txt"; rm -rf "$out"/pmc_FETCH_SIZE "$out"/pmc_WRITE_SIZE ;;
    sq)     (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES -d "$OLDPWD/$out/pmc_sq" -o p -- python "$OLDPWD/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$OLDPWD/$out/pmc_sq.log" 2>&1)
            python tools/rocprof_sq_summary.py "$(ls "$out"/pmc_sq/*results.db | head -1)" > "$out/sq_summary.txt" 2>&1; head -30 "$out/sq_summary.txt"; rm -rf "$out"/pmc_sq ;;
    modes)  timeout 1500 python tools/measure_modes.py "$out/overlap_modes.json" > "$out/modes.log" 2>&1; echo "modes exit $?"; tail -12 "$out/modes.log" ;;
    c3)     timeout 1700 python bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline > "$out/bench_config3.json" 2> "$out/bench_config3.err"; echo "c3 exit $?"
            python - "$out/bench_config3.json" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(" ms_per_step %.0f consensus %.0f overlap %.0f allocations %s pool_calls %s kernel_ms %s" % (d["ms_per_step"], d["consensus_ms_per_step"],
      d["overlap"]["ms_per_step"], d["allocations"], d["overlap"]["pool_calls"], d["kernel_ms"]))
P
            ;;
    parity) timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_k10.py tests/test_gpu_configs.py -m gpu -q -x --timeout=600 > "$out/pytest_parity.log" 2>&1; echo "parity exit $?"; tail -3 "$out/pytest_parity.log" ;;
    trace)  NDGPU_TRACE=1 NDGPU_PROF=1 timeout 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > "$out/bench_trace.json" 2> "$out/bench_trace.err"; echo "trace exit $?"; tail -c 300 "$out/bench_trace.json" ;;
    timeline) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/stats" -o s -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OLDPWD/$out/prof.log" 2>&1)
            db="$(ls "$out"/stats/*results.db | head -1)"
            python tools/rocprof_summary.py "$db" > "$out/kernel_stats.txt" 2>> "$out/prof.log"; head -30 "$out/kernel_stats.txt"
            python tools/rocprof_timeline.py "$db" > "$out/timeline.txt" 2>> "$out/prof.log"; rm -rf "$out/stats" ;;
    rccl)   # the collective leg of an N > 1 run with the one rank a one-GPU box allows: torch + RCCL + the product libraries in one process
            NDGPU_BENCH_FORCE_DIST=1 timeout ${RCCL_TIMEOUT:-200} python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
              bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_rccl_one_rank.json" 2> "$out/bench_rccl_one_rank.err"; echo "rccl exit $?"
            tail -c 400 "$out/bench_rccl_one_rank.err"; grep -o '"per_rank": [^]]*]' "$out/bench_rccl_one_rank.json" ;;
    cand)   # A/B of a kernel candidate (tools/kernel_candidate.py; CAND=<name>): the product library, then the candidate in its place with the parity leg
            timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$out/bench_product.json" 2> "$out/bench_product.err"; echo "product exit $?"
            timeout 200 python tools/kernel_candidate.py bench ${CAND:-snake32} --steps 10 --warmup 3 > "$out/bench_${CAND:-snake32}.json" 2> "$out/bench_${CAND:-snake32}.err"; echo "candidate exit $?"
            python - "$out/bench_product.json" "$out/bench_${CAND:-snake32}.json" <<'P'
import json, sys
for p in sys.argv[1:]:
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        print(" %s: ms_per_step %.1f cns %.1f ovl %.1f | kernel_ms %s | parity %s" % (p.split("/")[-1], d["ms_per_step"], d["consensus_ms_per_step"], d["overlap"]["ms_per_step"], {k: round(v) for k, v in d["kernel_ms"].items()}, d.get("parity")))
    except Exception as e:
        print(" %s: %r" % (p, e))
P
            ;;
    smoke)  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -3 ;;
    *) echo "unknown stage $s" ;;
  esac
  echo "   ($s: $(( $(date +%s) - t0 )) s)"
done
