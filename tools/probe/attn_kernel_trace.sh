#!/bin/bash
# Per-kernel averages of the attention core alone (tools/attn_time.py under rocprofv3 --kernel-trace --stats):
#   bash tools/probe/attn_kernel_trace.sh [tag] [attn_time.py args]       -> gpurun_out/<tag>attn_kernels.txt
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-}; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out"; rm -rf /tmp/prof_attn
rocprofv3 --kernel-trace --stats -d /tmp/prof_attn -o at --output-format csv -- python3 "$REPO/tools/attn_time.py" --iters 10 --warmup 2 "$@" > /tmp/prof_attn.log 2>&1
f=$(find /tmp/prof_attn -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' > "$REPO/gpurun_out/${TAG}attn_kernels.txt"
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: r["Name"]):
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
    if n.startswith("at::") or "elementwise" in n or "distribution" in n: continue
    print(f'{n[:60]:60s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
cat "$REPO/gpurun_out/${TAG}attn_kernels.txt"
