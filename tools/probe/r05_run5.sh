cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
bash tools/probe/attn_kernel_trace.sh r05a/n0_ --apply-mode 0 > /dev/null 2>&1
python tools/attn_time.py --apply-mode 0 2>&1 | grep stage
grep "partial\|apply\|combine" $O/n0_attn_kernels.txt
timeout 600 python -m pytest tests -x -q -m gpu -k "factor or attn or block" 2>&1 | tail -3
