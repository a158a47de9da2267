# interleaved A/B of two builds of the library on the bench step:  bash tools/probe/ab_lib.sh /path/to/variant.so [bench args...]
V=$1; shift
for i in 1 2 3; do
for v in "" "$V"; do
MDVIT_HIP_LIB=$v python bench.py --steps 10 --warmup 3 --no-extra-legs --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=[$v]', d['value'], 'img/s', d['ms_per_step'], 'ms', d['phase_ms'])"
done; done
