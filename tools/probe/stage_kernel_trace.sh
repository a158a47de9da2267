#!/bin/bash
# Per-kernel time table of ONE SerialBlock_adapt forward + backward at bs=32 (tools/block_roofline.py --eager under rocprofv3 --kernel-trace --stats):
#   bash tools/probe/stage_kernel_trace.sh [stages, default "2 3"] [tag]      -> gpurun_out/<tag>stage<k>_kernels.txt
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
STAGES=${1:-"2 3"}
TAG=${2:-}
cd /tmp && export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out"
for st in $STAGES; do
rm -rf /tmp/prof_st$st
rocprofv3 --kernel-trace --stats -d /tmp/prof_st$st -o st$st --output-format csv -- python3 "$REPO/tools/block_roofline.py" --batch 32 --stages $st --eager --iters 10 --warmup 3 > /dev/null 2>&1
f=$(find /tmp/prof_st$st -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' > "$REPO/gpurun_out/${TAG}stage${st}_kernels.txt"
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:45]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    print(f'{n[:100]:100s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f}%')
print("total ms", tot/1e6, "(13 iterations)")
PY
done
