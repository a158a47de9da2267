#!/bin/bash
# interleaved A/B of one environment switch on the default bench step:  bash tools/ab_env.sh VAR [rounds]
VAR=$1; R=${2:-2}
for i in $(seq $R); do
for v in 0 1; do
env $VAR=$v python bench.py --steps 10 --warmup 3 --no-extra-legs --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], 'img/s', d['ms_per_step'], 'ms', d['phase_ms'], 'host', d['host_enqueue_ms_per_step'], 'wait', d['host_throttle_wait_ms_per_step'])"
done; done
