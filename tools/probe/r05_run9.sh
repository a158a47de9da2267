cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
python tools/gemm_shapes_time.py profiles/r05_gemm_shapes_bs4.txt 2>&1 | grep -v amdgpu.ids | tee $O/gemm_shapes_alone.txt
