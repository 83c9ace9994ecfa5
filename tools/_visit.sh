bash tools/gpu_round.sh r06c
