cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
bash tools/ab_env.sh MDVIT_MLP_RC_BWD 3 2>&1 | tee $O/ab_mlp_rc_bwd_final.txt
