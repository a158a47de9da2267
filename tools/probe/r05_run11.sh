cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
python tools/probe/stem_wgrad_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O/stem_wgrad.txt
timeout 900 python -m pytest tests -x -q -m gpu -k "stem" 2>&1 | tail -3
