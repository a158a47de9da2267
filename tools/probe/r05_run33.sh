cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools/ab_env.sh MDVIT_GCONV2_TILES 3 2>&1 | tee $O/ab_gconv2_tiles.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace5 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra-legs --no-kernel-events > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
KS=$(find $O/trace5 -name "*kernel_stats.csv" | head -1)
python3 - $KS <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=[int(r['Calls']) for r in rows if 'adamw_kernel' in r['Name']][0]
for r in rows:
    n=r['Name']
    if any(k in n for k in ('gconv2','da_fwd')):
        print(f"{n.replace('(anonymous namespace)::','')[:70]:70s} {int(r['Calls'])/steps:5.1f}/step avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
rm -rf $O/trace5
