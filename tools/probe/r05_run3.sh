cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
python tools/mlp_rc_time.py 2>&1 | grep -v amdgpu.ids | tee $O/time_v2.txt
MDVIT_HIP_LIB=$PWD/mdvit_amd/lib/libmdvit_hip_r04.so python tools/mlp_rc_time.py --rounds 2 2>&1 | grep -v amdgpu.ids | tee $O/time_r04.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "mlp or dropout or drop" 2>&1 | tail -15
