#!/bin/bash
# Per-kernel breakdown of ONE SerialBlock_adapt forward + backward at bs=32 for each stage (eager launches under rocprofv3):
#     timeout 900 bash tools/block_trace.sh r02a      -> gpurun_out/<tag>_block_stage<s>_kernel_stats.csv
set -u
TAG=${1:-r02}
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export PYTHONPATH=$REPO
cd /tmp && export TMPDIR=/tmp
for S in 0 1 2 3; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/bt_$S" -o blk -- python3 "$REPO/tools/block_roofline.py" --eager --stages $S --iters 10 --warmup 2 \
        > "$OUT/${TAG}_block_stage${S}.txt" 2> /dev/null
    KS=$(ls /tmp/bt_$S/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$KS" ] && cp "$KS" "$OUT/${TAG}_block_stage${S}_kernel_stats.csv"
done
