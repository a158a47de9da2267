cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
timeout 2600 bash tools/profile_round.sh r05d > $O/profile_round.log 2>&1
python tools/mlp_rc_time.py --rounds 3 2>&1 | grep -v amdgpu | tee $O/mlp_rc_kernels_isolated.txt
python tools/mlp_rc_time.py --rounds 3 --planes 1 2>&1 | grep -v amdgpu | tee $O/mlp_rc_kernels_isolated_one_plane.txt
bash tools/probe/attn_kernel_trace.sh r05d/ > /dev/null 2>&1
python tools/attn_time.py 2>&1 | grep stage | tee $O/attn_time.txt
MDVIT_BENCH_GEMM_SHAPES=$O/gemm_shapes_bs4.txt python bench.py --steps 3 --warmup 2 --no-extra-legs --no-cpu-baseline > /dev/null 2>&1
python tools/gemm_shapes_time.py $O/gemm_shapes_bs4.txt 2>&1 | grep -v amdgpu.ids > $O/gemm_shapes_alone.txt
tail -3 $O/gemm_shapes_alone.txt
tail -5 $O/profile_round.log
