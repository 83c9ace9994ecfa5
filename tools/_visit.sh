timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streams.py tests/test_gpu_abi.py -q -x -m gpu -k "nms or step or stream or abi" 2>&1 | grep -E "passed|failed|Error" | tail -3
