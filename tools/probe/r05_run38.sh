cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 600 python -m pytest tests -x -q -m gpu -k "layernorm or ln_ or block_entry" 2>&1 | tail -3
for i in 1 2 3; do
for v in 64 0; do
MDVIT_LN_BWD_ROWS=$v python bench.py --steps 10 --warmup 3 --no-extra-legs --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LN_BWD_ROWS=$v', d['value'], 'img/s', d['ms_per_step'], 'ms', d['phase_ms'])"
done; done 2>&1 | tee $O/ab_ln_bwd_rows.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace7 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra-legs --no-kernel-events > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
KS=$(find $O/trace7 -name "*kernel_stats.csv" | head -1)
python3 - $KS <<'PY' | tee -a $O/ab_ln_bwd_rows.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=[int(r['Calls']) for r in rows if 'adamw_kernel' in r['Name']][0]
for r in rows:
    n=r['Name']
    if any(k in n for k in ('ln_bwd16','reduce_partials')):
        print(f"{n.replace('(anonymous namespace)::','')[:70]:70s} {int(r['Calls'])/steps:5.1f}/step avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
rm -rf $O/trace7
