bash tools/gpu_round.sh r06 2>&1 | tail -30
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06/pytest_gpu_full.log 2>&1; grep -E "passed|failed" gpurun_out/r06/pytest_gpu_full.log | tail -2
