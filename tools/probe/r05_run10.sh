cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
python tools/mlp_rc_time.py --rounds 3 --planes 1 2>&1 | grep -v amdgpu | tee $O/mlp_rc_kernels_isolated_one_plane.txt
timeout 900 python -m pytest tests -x -q -m gpu -k "bf16 or mlp" 2>&1 | tail -3
for v in 0 1; do MDVIT_MLP_RC_ONE_PLANE=$v python bench.py --precision bf16 --steps 10 --warmup 3 --no-extra-legs --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bs4 ONE_PLANE=$v', d['value'], d['ms_per_step'], d.get('drift_vs_parity_mode'))"; done
for v in 0 1; do MDVIT_MLP_RC_ONE_PLANE=$v python bench.py --precision bf16 --batch 16 --steps 5 --warmup 3 --no-extra-legs --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bs16 ONE_PLANE=$v', d['value'], d['ms_per_step'])"; done
