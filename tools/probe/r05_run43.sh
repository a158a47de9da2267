cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
for i in 1 2; do
for v in 0 1; do
MDVIT_EXP_TN_ONE_PLANE=$v python bench.py --batch 32 --steps 6 --warmup 2 --no-extra-legs --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bs32 MDVIT_EXP_TN_ONE_PLANE=$v', d['value'], 'img/s', d['ms_per_step'], 'ms', d['phase_ms'])"
done; done 2>&1 | tee $O/exp_tn_one_plane_bs32.txt
