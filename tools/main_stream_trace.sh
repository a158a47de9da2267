set -u
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o bench -- python3 "$REPO/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-extra-legs > $OUT/ms_under_rocprof.json 2> /tmp/tr.err
cd $REPO
LAST=$(python3 -c "import json,sys; print(3.0 * json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['ms_per_step'])" $OUT/ms_under_rocprof.json)
python3 tools/trace_summary.py /tmp/tr --steps 3 --last-ms "$LAST" --main-stream --top 400 > $OUT/ms_main_stream.txt
python3 tools/trace_summary.py /tmp/tr --steps 3 --last-ms "$LAST" --top 400 > $OUT/ms_all_streams.txt
head -5 $OUT/ms_main_stream.txt | cut -c1-160
