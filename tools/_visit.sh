python tools/_dbg_step.py test_nms_100k_properties 2>&1 | grep -c "num 2481"
for i in 1 2; do timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "one_launch_step or nms_step or multiscale_roi_align_boxes or nms" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4; done
