cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
for v in 2 3 4; do python tools/mlp_rc_time.py --tokens128 0 --variant $v 2>&1 | grep -v amdgpu.ids; done > $O/variants.txt
for l in abl1 abl2 abl3 abl4 abl6; do MDVIT_HIP_LIB=$PWD/mdvit_amd/lib/variants/libmdvit_hip_$l.so python tools/mlp_rc_time.py --tokens128 0 --rounds 2 2>&1 | grep -v amdgpu.ids; done > $O/ablations.txt
cat $O/variants.txt $O/ablations.txt
