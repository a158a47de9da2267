"""Print the per-shape GEMM table of a `bench.py --by-shape --detail out.json` run (ms/step, TFLOP/s, GB/s)."""
import json, sys
d = json.load(open(sys.argv[1]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms"])
tot = sum(r["ms"] for _, r in rows)
print(f"step {d['step_ms']:.1f} ms; GEMM events {tot/steps:.1f} ms/step")
for name, r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 50]:
    s = r["ms"] * 1e-3
    print(f"{r['ms']/steps:7.2f} ms/step {r['n']//steps:4d}x {r['ms']/r['n']*1e3:8.1f} us {r['flop']/s/1e12:6.1f} TF {r['bytes']/s/1e9:7.0f} GB/s  {name}")
