timeout 900 python -m pytest tests/test_gpu_roi_routes.py tests/test_gpu_baseline_sizes.py -q -x -m gpu -k "roi or multiscale or routes or fold" 2>&1 | tail -3
rm -rf _variants/*.o
for v in head new head new; do
  cp _variants/libtvmi_kernels_$v.so vision_amd/_lib/libtvmi_kernels.so
  for a in "14" "7 bf16" "14 bf16"; do python tools/roi_knock.py $v $a 2>&1 | tail -1; done
  python tools/nhwc_timing.py 2>&1 | grep -E "order 1|nhwc bf16|bits"
done
cp _variants/libtvmi_kernels_new.so vision_amd/_lib/libtvmi_kernels.so
