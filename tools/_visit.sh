mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/r06/pytest_gpu_full.log 2>&1; tail -5 gpurun_out/r06/pytest_gpu_full.log
timeout 900 python bench.py > gpurun_out/r06/bench_b.json 2> gpurun_out/r06/bench_b.err; cut -c1-300 gpurun_out/r06/bench_b.json
