cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do
for l in variants/libmdvit_hip_pre_tn.so libmdvit_hip.so; do
MDVIT_HIP_LIB=$PWD/mdvit_amd/lib/$l python bench.py --steps 10 --warmup 3 --no-extra-legs --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', d['value'], 'img/s', d['ms_per_step'], 'ms', d['phase_ms'])"
done; done 2>&1 | tee $O/ab_tn_joint.txt
