cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_like_bench.json 2> $O/driver_like_bench.err
echo "bench wall seconds: $(( $(date +%s) - T0 ))"
python -c "
import json
d=json.loads(open('$O/driver_like_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['phase_ms'], 'host', d['host_enqueue_ms_per_step'], 'roofline', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'])
print({k:(v.get('value') if isinstance(v,dict) else None) for k,v in d.items() if k in ('bs32','bf16_speed_mode','bs16_bf16','transfuse_bs8','base_bs4_gpu')})
"
