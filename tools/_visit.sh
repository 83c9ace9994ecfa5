mkdir -p gpurun_out/r06d
( time timeout 900 python bench.py > gpurun_out/r06d/bench.json 2> gpurun_out/r06d/bench.err ) 2>&1 | tail -4; echo "rc=$?"; wc -c gpurun_out/r06d/bench.json; tail -5 gpurun_out/r06d/bench.err | cut -c1-300
