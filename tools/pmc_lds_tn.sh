#!/bin/bash
# LDS bank-conflict counters of the transposing-read wgrad kernel (and the implicit-conv / NT kernels that share the run):
#     bash tools/pmc_lds_tn.sh   -> gpurun_out/r02_pmc_lds_tn.txt
set -u
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_lds -o pmc -- python3 "$REPO/tools/tn_check.py" --time > /tmp/pmc_lds.log 2>&1
cd "$REPO"
python3 - <<'PY' > "$OUT/r02_pmc_lds_tn.txt"
import csv, glob, collections
f = sorted(glob.glob('/tmp/pmc_lds/**/*counter_collection.csv', recursive=True))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in f:
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_LDS_IDX_ACTIVE': n[k] += 1
print("kernel | launches | SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra LDS cycles per active LDS cycle)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', 0))[:14]:
    a = v.get('SQ_LDS_IDX_ACTIVE', 0.0); c = v.get('SQ_LDS_BANK_CONFLICT', 0.0)
    print(f"{k[:70]:70s} {n[k]:6d}  {c:.3e} / {a:.3e} = {c / a if a else 0:.4f}")
PY
cat "$OUT/r02_pmc_lds_tn.txt"
