cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
python tools/probe/stem_fwd_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O/stem_fwd.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
