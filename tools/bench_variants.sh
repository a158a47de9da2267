run() { printf "%-44s" "$*"; timeout 250 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra-legs --no-kernel-events "$@" 2>&1 | tail -1 | python3 -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'))
except Exception as e: print('FAILED', e)"; }
run --model mdvit_dsn
run --decoder MLP
run --decoder DeepLabV3
run --decoder Transformer
run --model base
run --reference-sweeps
run --no-side-stream
run --precision fp32
run --precision bf16
run --graph
run --host-inputs
run --torch-adamw
run --batch 8
run --fuse-images 4
