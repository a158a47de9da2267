cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
MDVIT_BENCH_GEMM_SHAPES=$O/gemm_shapes_bs4.txt python bench.py --steps 10 --warmup 3 --no-extra-legs --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bs4', d['value'], d['ms_per_step'], d['roofline']['frac'])"
head -70 $O/gemm_shapes_bs4.txt
