#!/bin/bash
# Where the cycles of the weight-gradient kernel go (SQ counters, two passes):   bash tools/pmc_tn_stalls.sh [M N K]   -> gpurun_out/pmc_tn_stalls.txt
set -u
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_tn_a -o pmc -- python3 "$REPO/tools/probe/tn_one_shape.py" "$@" > /tmp/pmc_tn_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA --output-format csv -d /tmp/pmc_tn_b -o pmc -- python3 "$REPO/tools/probe/tn_one_shape.py" "$@" > /tmp/pmc_tn_b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES --output-format csv -d /tmp/pmc_tn_c -o pmc -- python3 "$REPO/tools/probe/tn_one_shape.py" "$@" > /tmp/pmc_tn_c.log 2>&1
cd "$REPO"
python3 - <<'PY' > "$OUT/pmc_tn_stalls.txt"
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for d in ("/tmp/pmc_tn_a", "/tmp/pmc_tn_b", "/tmp/pmc_tn_c"):
    for path in sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True)):
        for r in csv.DictReader(open(path)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k, v in agg.items():
    if 'gemm_tn' not in k:
        continue
    wc = v.get('SQ_WAVE_CYCLES', 0.0) / max(n[k]['SQ_WAVE_CYCLES'], 1)
    print(k, "(per launch; % of SQ_WAVE_CYCLES = cycles summed over resident waves)")
    for c in sorted(v):
        x = v[c] / max(n[k][c], 1)
        print(f"   {c:28s} {x:12.4g}  {100 * x / wc if wc else 0:6.1f} %")
PY
cat "$OUT/pmc_tn_stalls.txt"
