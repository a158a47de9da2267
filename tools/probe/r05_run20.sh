cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
python3 tools/block_roofline.py --batch 32 --json $O/block_roofline_bs32.json 2>/dev/null | head -8
MDVIT_PM_GEMM=0 python3 tools/block_roofline.py --batch 32 2>/dev/null | head -5
