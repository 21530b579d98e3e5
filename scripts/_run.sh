scripts/gpu_profile.sh r01g
