mkdir -p gpurun_out/r06e
( for sd in 31 32 33 34; do timeout 300 python tools/fuzz_gpu.py $sd 2>&1 | tail -2; done; timeout 400 python tools/step_stress.py 90 2>&1 | tail -2 ) > gpurun_out/r06e/fuzz.log 2>&1; cat gpurun_out/r06e/fuzz.log | cut -c1-300
timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | tail -15 > gpurun_out/r06e/pytest_gpu.log; tail -4 gpurun_out/r06e/pytest_gpu.log | cut -c1-200
