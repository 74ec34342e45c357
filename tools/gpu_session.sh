#!/bin/bash
# One GPU-box session, in stages, so that a round's metered GPU minutes go to measurements in a fixed order and every stage leaves
# its evidence under gpurun_out/ even if a later one is cut off.  Run through gpurun from the repo root:
#     gpurun --timeout 2400 -- 'bash tools/gpu_session.sh r03 tests bench prof pmc'
# stages:  tests  pytest -m gpu (the new files test_zz_gpu_* included)           -> gpurun_out/<tag>/pytest.log
#          bench  python bench.py (default config), then --config 3 if asked     -> gpurun_out/<tag>/bench*.json
#          prof   rocprofv3 --kernel-trace --stats of the default bench          -> gpurun_out/<tag>/stats/ + kernel_stats.txt
#          pmc    FETCH_SIZE / WRITE_SIZE passes (no trace flags with --pmc)     -> gpurun_out/<tag>/pmc_*/
# Copy what is to be judged into profiles/ afterwards (gpurun_out/ is scratch).
set -u
tag=${1:-session}; shift || true
stages=${*:-tests bench}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
for s in $stages; do
  echo "== stage $s =="
  case $s in
    tests)
      timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > "$out/pytest.log" 2>&1
      echo "pytest exit $?" | tee -a "$out/pytest.log"; tail -5 "$out/pytest.log" ;;
    bench)
      timeout 900 python bench.py --steps 3 --warmup 1 > "$out/bench_default.json" 2> "$out/bench_default.err"
      echo "bench exit $?"; tail -c 1500 "$out/bench_default.json" ;;
    bench3)
      timeout 1800 python bench.py --config 3 --steps 1 --warmup 1 > "$out/bench_config3.json" 2> "$out/bench_config3.err"
      echo "bench3 exit $?"; tail -c 1500 "$out/bench_config3.json" ;;
    prof)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/stats" -o s -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 > "$OLDPWD/$out/prof.log" 2>&1)
      python tools/rocprof_summary.py "$(ls "$out"/stats/*results.db | head -1)" > "$out/kernel_stats.txt" 2>> "$out/prof.log"; head -25 "$out/kernel_stats.txt" ;;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c -d "$OLDPWD/$out/pmc_$c" -o p -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 > "$OLDPWD/$out/pmc_$c.log" 2>&1)
      done
      python tools/rocprof_pmc_summary.py "$(ls "$out"/pmc_FETCH_SIZE/*results.db | head -1)" "$(ls "$out"/pmc_WRITE_SIZE/*results.db | head -1)" > "$out/pmc_summary.txt" 2>&1; head -30 "$out/pmc_summary.txt" ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ;;
    *) echo "unknown stage $s" ;;
  esac
done
