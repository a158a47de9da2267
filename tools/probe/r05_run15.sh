cd $GRAFT_REPO_ROOT
REPO=$PWD; O=$REPO/gpurun_out/r05e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra-legs --no-kernel-events > $O/under_rocprof.json 2> $O/trace.err
cd $REPO
MS=$(python3 -c "import json,sys; print(json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['ms_per_step'])" $O/under_rocprof.json)
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/stream_timeline.py $T --last-ms $MS --bucket-ms 1 > $O/timeline.txt
python3 tools/main_gaps.py $T --last-ms $MS --min-us 40 --top 60 > $O/main_gaps.txt
rm -rf $O/trace
echo "ms per step under rocprof: $MS"
cat $O/timeline.txt | head -60
cat $O/main_gaps.txt | head -90
