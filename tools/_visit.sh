mkdir -p gpurun_out/r06
timeout 400 python tools/step_stress.py 240 2>&1 | tail -2 | tee gpurun_out/r06/step_stress.log
timeout 900 python tools/fuzz_gpu.py 2>&1 | tail -3 | tee gpurun_out/r06/fuzz_tail.log
