mkdir -p gpurun_out/r4a
timeout 1500 python -m pytest tests -x -q -m gpu -s > gpurun_out/r4a/pytest.log 2>&1; echo pytest_rc=$?; grep "drop-in\|passed\|failed\|Error" gpurun_out/r4a/pytest.log | tail -12
