cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
MDVIT_STEM_FWD32=0 python tools/probe/stem_fwd_time.py 2>&1 | grep -v amdgpu.ids | tee $O/stem_fwd.txt
python tools/probe/stem_fwd_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O/stem_fwd.txt
timeout 900 python -m pytest tests -x -q -m gpu -k "stem or conv or golden" 2>&1 | tail -3
for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-extra-legs --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bs4', d['value'], d['ms_per_step'], d['phase_ms'])"; done
