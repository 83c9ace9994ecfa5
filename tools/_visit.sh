mkdir -p gpurun_out/r06
(timeout 200 python tools/step_stress.py 90 2>&1 | tail -3) > gpurun_out/r06/stress.log; cat gpurun_out/r06/stress.log
for seed in 21 22 23; do (timeout 200 python tools/fuzz_gpu.py $seed 2>&1 | tail -3) >> gpurun_out/r06/fuzz.log; done; cat gpurun_out/r06/fuzz.log
