cd $GRAFT_REPO_ROOT
bash tools/pmc_stalls.sh gemm_pm r05e_gemm_pm_pmc_stalls.txt tools/probe/gemm_pm_one_shape.py 16384 320 1280 20
bash tools/pmc_stalls.sh mlp_rc r05e_mlp_rc_pmc_stalls.txt tools/mlp_rc_time.py --rounds 1 --tokens128 0
cat gpurun_out/r05e_gemm_pm_pmc_stalls.txt | cut -c1-160
grep -A28 "mlp_rc_fwd3_kernel<64, 4, true" gpurun_out/r05e_mlp_rc_pmc_stalls.txt | head -40 | cut -c1-160
