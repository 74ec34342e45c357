mkdir -p gpurun_out/r4b
timeout 600 python -m pytest tests/test_gpu_robust.py -x -q > gpurun_out/r4b/pytest.log 2>&1; echo pytest_rc=$?; tail -5 gpurun_out/r4b/pytest.log
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4b/stats -o s -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r4b/bench_prof.json 2> $R/gpurun_out/r4b/bench_prof.err
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/r4b/pmc_$c -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r4b/pmc_$c.json 2> $R/gpurun_out/r4b/pmc_$c.err; done
cd $R
for f in $(find gpurun_out/r4b/stats -name "*.db"); do python tools/rocprof_summary.py $f > gpurun_out/r4b/kernel_stats.txt; python tools/rocprof_timeline.py $f 0.5 1.0 > gpurun_out/r4b/timeline.txt; done
python tools/rocprof_pmc_summary.py $(find gpurun_out/r4b/pmc_FETCH_SIZE -name "*.db" | head -1) $(find gpurun_out/r4b/pmc_WRITE_SIZE -name "*.db" | head -1) > gpurun_out/r4b/pmc_summary.txt 2>&1
find gpurun_out/r4b -name "*.db" -delete
head -12 gpurun_out/r4b/kernel_stats.txt; head -20 gpurun_out/r4b/pmc_summary.txt
