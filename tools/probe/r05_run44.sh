mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r05e
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tr_bf16
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_bf16 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --precision bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-extra-legs --no-kernel-events > /tmp/tr_bf16.log 2>&1
cd $GRAFT_REPO_ROOT
KS=$(find /tmp/tr_bf16 -name "*kernel_stats.csv" | head -1)
python3 - $KS <<'PY' | tee gpurun_out/r05e/bf16_mode_kernels.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=[int(r['Calls']) for r in rows if 'adamw_kernel' in r['Name']][0]
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('bf16 mode, bs=4: steps',steps,'launches/step',sum(int(r['Calls']) for r in rows)/steps,'kernel ms/step',tot/steps/1e6)
for r in rows[:40]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.2f}% {int(r['Calls'])/steps:6.1f}/step avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:110]}")
PY
