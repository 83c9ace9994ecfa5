echo "scratch script of single GPU visits (see tools/gpu_round.sh)"
