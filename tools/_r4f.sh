mkdir -p gpurun_out/r4f
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r4f/pytest.log 2>&1; echo pytest_rc=$?; tail -4 gpurun_out/r4f/pytest.log
