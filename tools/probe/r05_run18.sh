cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 300 python tools/gemm_pm_check.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_pm_check.txt | grep -v "^  ok" | tail -60
