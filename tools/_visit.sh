python tools/dcn_fwd_timing.py dcn.xcd_tiles=1 dcn.xcd_tiles=1 dcn.xcd_tiles=1 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py tests/test_gpu_abi.py -q -x -m gpu -k "deform and not backward" 2>&1 | tail -3
