cd $GRAFT_REPO_ROOT
REPO=$PWD; O=$REPO/gpurun_out/r05e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra-legs --no-kernel-events > $O/under_rocprof.json 2> $O/trace.err
cd $REPO
LAST=$(python3 -c "import json,sys; print(3.0 * json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['ms_per_step'])" $O/under_rocprof.json)
python3 tools/trace_summary.py $O/trace --steps 3 --last-ms $LAST --forward --top 70 > $O/forward_summary.txt
python3 tools/trace_summary.py $O/trace --steps 3 --last-ms $LAST --top 40 > $O/step_summary.txt
rm -rf $O/trace
cat $O/forward_summary.txt
