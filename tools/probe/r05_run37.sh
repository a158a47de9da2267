cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
bash tools/ab_env.sh MDVIT_EXP_SKIP_AUX_SWEEP 2 2>&1 | tee $O/ab_exp_skip_aux_sweep.txt
