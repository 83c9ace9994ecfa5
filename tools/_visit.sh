echo "see tools/gpu_round.sh; this file is the scratch script of single GPU visits"
