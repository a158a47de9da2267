#!/bin/bash
# interleaved A/B/C... of several VALUES of one environment variable on the default bench step:  bash tools/ab_values.sh VAR "v1 v2 v3" [rounds]
VAR=$1; VALS=$2; R=${3:-2}
for i in $(seq $R); do
for v in $VALS; do
env $VAR=$v python bench.py --steps 12 --warmup 4 --no-extra-legs --no-cpu-baseline --settle-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], 'img/s', d['ms_per_step'], 'ms', 'host', d['host_enqueue_ms_per_step'])"
done; done
