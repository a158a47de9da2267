cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 600 python -m pytest tests -x -q -m gpu -k "attn or factor or block_entry" 2>&1 | tail -2
for v in 0 1; do
export MDVIT_FA_APPLY3_XCD=$v
bash tools/probe/attn_kernel_trace.sh r05e/xcd${v}_ > /dev/null 2>&1
echo "== MDVIT_FA_APPLY3_XCD=$v (32 images per stage: tools/attn_time.py under rocprofv3)"; grep "fa_bwd_apply3" $O/xcd${v}_attn_kernels.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmcx
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcx -o pmc -- python3 $GRAFT_REPO_ROOT/tools/attn_time.py --iters 3 --warmup 1 > /tmp/pmcx.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/pmcx/**/*counter_collection.csv', recursive=True)
acc=collections.defaultdict(lambda:[0,0.0])
for fn in f:
    for r in csv.DictReader(open(fn)):
        if r.get('Counter_Name')=='FETCH_SIZE' and 'fa_bwd_apply3' in r['Kernel_Name']:
            k=r['Kernel_Name'].split('(')[0][-28:]+' grid '+r.get('Grid_Size','?')
            acc[k][0]+=1; acc[k][1]+=float(r['Counter_Value'])
for k,(n,v) in sorted(acc.items()):
    print(f"   {k}: {n} launches, FETCH_SIZE {v/n*2*1024/1e6:8.1f} MB per launch (x2 corrected)")
PY
done 2>&1 | tee $O/fa_apply3_xcd.txt
unset MDVIT_FA_APPLY3_XCD
bash tools/ab_env.sh MDVIT_FA_APPLY3_XCD 3 2>&1 | tee $O/ab_fa_apply3_xcd.txt
