"""Superseded by tools/pmc_stalls.sh (round 4), which prints every counter with its unit: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are QUAD-cycles summed
over the resident waves, SQ_VALU_MFMA_BUSY_CYCLES is cycles -- this script divided the one by the other and overstated the MFMA share four times.  Kept for
reading old counter directories: MFMA busy is now reported as SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES)."""
import csv, glob, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        n = r["Kernel_Name"]
        if "gemm_f32_kernel" not in n and "mlp_" not in n and "rc_" not in n and "gemm_" not in n: continue
        n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", "")) + f" grid={r.get('Grid_Size','?')}"
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
for n, c in agg.items():
    w = c.get("SQ_WAVE_CYCLES", 1.0)
    print(n)
    for k, v in sorted(c.items()):
        if k == "SQ_VALU_MFMA_BUSY_CYCLES":
            print(f"   {k}={v:.3g} cycles  (per wave: {100 * v / (4 * w):.1f} % of its life = / (4 x SQ_WAVE_CYCLES))")
        elif k.startswith("SQ_INSTS") or k == "SQ_WAVES":
            print(f"   {k}={v:.3g}")
        else:
            print(f"   {k}={v:.3g} quad-cycles ({100 * v / w:.0f} % of SQ_WAVE_CYCLES)")
