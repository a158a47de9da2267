cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
bash tools/ab_env.sh MDVIT_EXP_TN_ONE_PLANE 3 2>&1 | tee $O/ab_exp_tn_one_plane.txt
