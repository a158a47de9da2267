cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
export MDVIT_HIP_LIB=$PWD/mdvit_amd/lib/libmdvit_hip_r04.so
bash tools/probe/attn_kernel_trace.sh r05a/r04_ > /dev/null 2>&1
python tools/attn_time.py 2>&1 | grep stage > $O/attn_time_r04.txt
unset MDVIT_HIP_LIB
for m in 0 1 2; do
  bash tools/probe/attn_kernel_trace.sh r05a/m${m}_ --apply-mode $m > /dev/null 2>&1
  python tools/attn_time.py --apply-mode $m 2>&1 | grep stage > $O/attn_time_m$m.txt
done
for f in r04 m0 m1 m2; do echo "== $f"; cat $O/attn_time_$f.txt; done
paste $O/r04_attn_kernels.txt $O/m1_attn_kernels.txt | awk '{printf "%-40s %8s %8s\n", $1, $5, $11}'
echo; grep "apply" $O/m0_attn_kernels.txt $O/m2_attn_kernels.txt
timeout 600 python -m pytest tests -x -q -m gpu -k "factor or attn or block" 2>&1 | tail -3
