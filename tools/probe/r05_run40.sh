cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "128_row or few_tile or 256_tile" 2>&1 | tail -5
python tools/gemm_pm_check.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/gemm_pm_splits.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools/ab_env.sh MDVIT_PM_SPLIT 3 2>&1 | tee $O/ab_pm_split.txt
