cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
python tools/probe/conv_bridge_plans.py 2>&1 | grep -v amdgpu.ids | tee $O/conv_bridge_plans.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools/ab_env.sh MDVIT_CONV_SPLIT_PLAN 3 2>&1 | tee $O/ab_conv_split_plan.txt
