#!/bin/bash
# The measurement set committed under profiles/ for one round.  Run on the GPU box from the repo root:
#     timeout 1500 bash tools/profile_round.sh r02a
# writes gpurun_out/<tag>_*: the default bench line, a rocprofv3 kernel trace of the same command (+ per-kernel summary and
# idle-gap analysis), two --pmc passes (FETCH_SIZE, WRITE_SIZE -- counters in their own runs, never with tracing domains)
# folded into HBM bytes per launch, and the MHSA+DA block roofline at bs=32.
set -u
TAG=${1:-r01}
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export PYTHONPATH=$REPO
python3 bench.py > "$OUT/${TAG}_bench_bs4.json" 2> "$OUT/${TAG}_bench.err"
python3 tools/block_roofline.py --batch 32 --json "$OUT/${TAG}_block_roofline_bs32.json" > "$OUT/${TAG}_block_roofline_bs32.txt" 2> /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -o bench -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline \
    > "$OUT/${TAG}_bench_bs4_under_rocprof.json" 2> "$OUT/${TAG}_trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_pmc_fetch" -o pmc -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events \
    > "$OUT/${TAG}_pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_pmc_write" -o pmc -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events \
    > "$OUT/${TAG}_pmc_write.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_tf_trace" -o bench -- python3 "$REPO/bench.py" --model transfuse --batch 8 --size 256 --steps 3 --warmup 1 \
    --no-cpu-baseline --no-kernel-events > "$OUT/${TAG}_transfuse_bs8_under_rocprof.json" 2> "$OUT/${TAG}_tf_trace.err"
cd "$REPO"
python3 bench.py --model transfuse --batch 8 --size 256 --no-cpu-baseline > "$OUT/${TAG}_transfuse_bs8.json" 2>> "$OUT/${TAG}_bench.err"
TF=$(ls "$OUT/${TAG}_tf_trace"/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$TF" ] && cp "$TF" "$OUT/${TAG}_transfuse_bs8_kernel_stats.csv"
rm -rf "$OUT/${TAG}_tf_trace"
# the last three (timed) steps of the profiled run: 3 x its own ms_per_step back from the end of the trace
LAST=$(python3 -c "import json,sys; print(3.0 * json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['ms_per_step'])" "$OUT/${TAG}_bench_bs4_under_rocprof.json")
python3 tools/trace_summary.py "$OUT/${TAG}_trace" --steps 3 --last-ms "$LAST" > "$OUT/${TAG}_bench_bs4_trace_summary.txt"
python3 tools/trace_summary.py "$OUT/${TAG}_trace" --steps 3 --last-ms "$LAST" --main-stream > "$OUT/${TAG}_bench_bs4_main_stream.txt"
python3 tools/trace_gaps.py "$OUT/${TAG}_trace" --last-ms "$LAST" > "$OUT/${TAG}_bench_bs4_idle_gaps.txt"
KS=$(ls "$OUT/${TAG}_trace"/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$KS" ] && cp "$KS" "$OUT/${TAG}_bench_bs4_kernel_stats.csv"
python3 tools/pmc_summary.py "$OUT/${TAG}_pmc_fetch" "$OUT/${TAG}_pmc_write" "$OUT/${TAG}_pmc_traffic.json" > "$OUT/${TAG}_pmc_top.txt"
rm -rf "$OUT/${TAG}_trace" "$OUT/${TAG}_pmc_fetch" "$OUT/${TAG}_pmc_write"       # raw traces stay on the box (tens of MB)
tail -1 "$OUT/${TAG}_bench_bs4.json" | cut -c1-300
cat "$OUT/${TAG}_block_roofline_bs32.txt"
head -12 "$OUT/${TAG}_bench_bs4_trace_summary.txt"
