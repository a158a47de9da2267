# interleaved A/B of the weight-gradient kernel's workgroup-count rule in the bench step (0: 384 in total, -1: whole multiples of the CU count): bash tools/probe/ab_tn_workgroups.sh
for i in 1 2 3; do
for v in 0 -1; do
MDVIT_TN_WORKGROUPS=$v python bench.py --steps 10 --warmup 3 --no-extra-legs --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MDVIT_TN_WORKGROUPS=$v', d['value'], 'img/s', d['ms_per_step'], 'ms', d['phase_ms'], 'host', d['host_enqueue_ms_per_step'])"
done; done
