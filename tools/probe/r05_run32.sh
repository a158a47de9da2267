cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "adapter or factor_att" 2>&1 | tail -5
