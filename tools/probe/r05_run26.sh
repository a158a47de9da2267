cd $GRAFT_REPO_ROOT
python tools/probe/gemm_pm_tiled_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05e/gemm_pm_tiled_probe.txt
