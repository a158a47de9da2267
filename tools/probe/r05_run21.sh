cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools/ab_env.sh MDVIT_UPSAMPLE_BWD_LDS 3 2>&1 | tee $O/ab_upsample_lds.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace2 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra-legs --no-kernel-events > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
KS=$(find $O/trace2 -name "*kernel_stats.csv" | head -1)
grep -E "upsample_multi|da_fwd|splitk_reduce|reduce_partials" $KS | awk -F, '{printf "%-80s calls %6d avg %8.1f us\n", substr($1,1,80), $2, $4/1e3}'
rm -rf $O/trace2
