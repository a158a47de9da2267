cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
{
python tools/probe/step_determinism.py 16 1 bs4 2>&1 | grep -v amdgpu.ids | tail -8
python tools/probe/step_determinism.py 6 1 bs32 2>&1 | grep -v amdgpu.ids | tail -5
python tools/probe/step_determinism.py 8 1 transfuse 2>&1 | grep -v amdgpu.ids | tail -5
for i in 1 2 3; do timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -1; done
} 2>&1 | tee $O/soak_final_tree.txt
