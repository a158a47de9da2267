cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_block.py -x -q -m gpu -k "adapter or block_entry" 2>&1 | tail -5
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools/ab_env.sh MDVIT_DA_MANY_BWD 3 2>&1 | tee $O/ab_da_many_bwd.txt
