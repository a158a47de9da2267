cd $GRAFT_REPO_ROOT
REPO=$PWD; O=$REPO/gpurun_out/r05e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra-legs --no-kernel-events > $O/under_rocprof.json 2> $O/trace.err
cd $REPO
MS=$(python3 -c "import json,sys; print(3*json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['ms_per_step'])" $O/under_rocprof.json)
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
for r in 1 2; do python3 tools/trace_summary.py $T --steps 3 --last-ms $MS --stream-rank $r --top 45 > $O/stream_rank$r.txt; done
rm -rf $O/trace
cat $O/stream_rank1.txt $O/stream_rank2.txt
