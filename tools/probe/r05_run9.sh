cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
python tools/mlp_rc_time.py --rounds 3 2>&1 | grep -v amdgpu > $O/mlp_rc_kernels_isolated.txt
python tools/mlp_rc_time.py --tokens64 0 --variant 32 --rounds 3 2>&1 | grep -v amdgpu > $O/mlp_rc_c128_variant32.txt
cat $O/mlp_rc_kernels_isolated.txt $O/mlp_rc_c128_variant32.txt
timeout 900 python -m pytest tests -x -q -m gpu -k "mlp or block or dropout" 2>&1 | tail -3
python tools/block_roofline.py --batch 32 --stages 0 2>&1 | grep -v amdgpu | tail -9
