# whole-step HBM traffic of the bs=32 step (BASELINE configs[2]'s per-GPU batch): two rocprofv3 --pmc passes (counters only), summed over every dispatch of the sampled steps
REPO=$GRAFT_REPO_ROOT
O=$REPO/gpurun_out/r05e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc32f /tmp/pmc32w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc32f -o pmc -- python3 $REPO/bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-extra-legs > /tmp/pmc32f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc32w -o pmc -- python3 $REPO/bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-extra-legs > /tmp/pmc32w.log 2>&1
tail -1 /tmp/pmc32f.log | cut -c1-300
cd $REPO
python3 tools/pmc_summary.py /tmp/pmc32f /tmp/pmc32w $O/pmc_traffic_bs32.json > $O/pmc_top_bs32.txt
python3 - <<'PY' | tee -a $O/pmc_top_bs32.txt
import json,os
d=json.load(open(os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out/r05e/pmc_traffic_bs32.json')))['kernels']
steps=d['adamw_kernel']['launches_sampled']
tot=sum(v['hbm_bytes_per_launch']*v['launches_sampled'] for v in d.values())/steps
rd=sum(2*v['fetch_kib_raw_per_launch']*1024*v['launches_sampled'] for v in d.values())/steps
print(f"bs=32 step: {steps} steps sampled, {tot/1e9:.1f} GB of HBM traffic per step ({rd/1e9:.1f} read, {(tot-rd)/1e9:.1f} written), {sum(v['launches_sampled'] for v in d.values())/steps:.0f} launches per step")
PY
