"""Repeat one GPU test function N times in one process (flake hunting):  python tools/probe/flake_loop.py tests.test_gpu_model test_aux_sweep_on_its_own_stream_equals_single_stream_step 10"""
import importlib, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mod, fn, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
m = importlib.import_module(mod)
bad = 0
for i in range(n):
    try:
        getattr(m, fn)()
    except AssertionError as e:
        bad += 1
        print(f"run {i}: FAIL {str(e)[:160]}", flush=True)
print(f"{fn}: {bad} failures in {n} runs")
