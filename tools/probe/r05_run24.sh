cd $GRAFT_REPO_ROOT
REPO=$PWD; O=$REPO/gpurun_out/r05e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace3 -o bench -- python3 $REPO/bench.py --precision bf16 --batch 16 --steps 3 --warmup 2 --no-cpu-baseline --no-extra-legs --no-kernel-events > $O/bf16_bs16_under_rocprof.json 2> $O/trace3.err
cd $REPO
MS=$(python3 -c "import json,sys; print(3*json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['ms_per_step'])" $O/bf16_bs16_under_rocprof.json)
T=$(find $O/trace3 -name "*kernel_trace.csv" | head -1)
python3 tools/trace_summary.py $T --steps 3 --last-ms $MS --top 45 > $O/bf16_bs16_trace_summary.txt
rm -rf $O/trace3
cat $O/bf16_bs16_trace_summary.txt | cut -c1-150
