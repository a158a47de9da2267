set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
O=gpurun_out/r05a
MDVIT_HIP_LIB=$PWD/mdvit_amd/lib/libmdvit_hip_r04.so python tools/mlp_rc_time.py --save $O/old.pt > $O/time_old.txt 2>&1
python tools/mlp_rc_time.py --save $O/new.pt > $O/time_new.txt 2>&1
MDVIT_HIP_LIB=$PWD/mdvit_amd/lib/libmdvit_hip_r04.so python tools/mlp_rc_time.py > $O/time_old2.txt 2>&1
python tools/mlp_rc_time.py > $O/time_new2.txt 2>&1
python tools/compare_saved.py $O/old.pt $O/new.pt > $O/compare.txt 2>&1
rm -f $O/old.pt $O/new.pt
cat $O/time_old.txt $O/time_new.txt $O/time_old2.txt $O/time_new2.txt $O/compare.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "mlp or dropout or drop" 2>&1 | tail -15
