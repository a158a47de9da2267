import csv, glob, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        n = r["Kernel_Name"]
        if "gemm_f32_kernel" not in n and "mlp_" not in n and "rc_" not in n: continue
        n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", "")) + f" grid={r.get('Grid_Size','?')}"
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
for n, c in agg.items():
    w = c.get("SQ_WAVE_CYCLES", 1.0)
    print(n)
    print("   " + "  ".join(f"{k}={v:.3g} ({100*v/w:.0f}%)" for k, v in sorted(c.items())))
