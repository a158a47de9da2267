#!/bin/bash
# Where the cycles of ONE kernel go (SQ counters, four --pmc passes, each in its own run, no tracing domains):
#     bash tools/pmc_stalls.sh <kernel-name substring> <out file under gpurun_out/> <python script> [args...]
# UNITS (MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units"): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count QUAD-cycles summed over
# the resident waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles (= 32 x SQ_INSTS_MFMA for v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE = the launch's
# duration in shader cycles at the clock the chip actually ran, SUMMED over the 8 XCDs.  Printed:
#   * the wait / issue buckets as a share of SQ_WAVE_CYCLES (quad-cycles against quad-cycles),
#   * "MFMA busy, per wave"  = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES): the share of a wave's life with one of ITS MFMAs in the pipe,
#   * "MFMA pipe utilisation" = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): what the matrix pipes of the chip did during the launch
#     (rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs: 4.68e6 for a 290 us launch = 8 x 585k cycles at 2.02 GHz).
set -u
PAT=$1; NAME=$2; SCRIPT=$3; shift 3
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcs_a /tmp/pmcs_b /tmp/pmcs_c /tmp/pmcs_d
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmcs_a -o pmc -- python3 "$REPO/$SCRIPT" "$@" > /tmp/pmcs_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA --output-format csv -d /tmp/pmcs_b -o pmc -- python3 "$REPO/$SCRIPT" "$@" > /tmp/pmcs_b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmcs_c -o pmc -- python3 "$REPO/$SCRIPT" "$@" > /tmp/pmcs_c.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcs_d -o pmc -- python3 "$REPO/$SCRIPT" "$@" > /tmp/pmcs_d.log 2>&1
cd "$REPO"
python3 - "$PAT" <<'PY' > "$OUT/$NAME"
import csv, glob, collections, sys
pat = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for d in ("/tmp/pmcs_a", "/tmp/pmcs_b", "/tmp/pmcs_c", "/tmp/pmcs_d"):
    for path in sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True)):
        for r in csv.DictReader(open(path)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k, v in agg.items():
    if pat not in k:
        continue
    a = {c: v[c] / max(n[k][c], 1) for c in v}
    wc = a.get('SQ_WAVE_CYCLES', 0.0)
    print(k, f"(per launch, {max(n[k].values())} launches)")
    for c in sorted(a):
        unit = "cycles" if c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE") else ("count" if c.startswith("SQ_INSTS") or c == "SQ_WAVES" else "quad-cycles")
        share = f"{100 * a[c] / wc:6.1f} % of SQ_WAVE_CYCLES" if unit == "quad-cycles" and wc else ""
        print(f"   {c:28s} {a[c]:12.4g} {unit:12s} {share}")
    mf, gui, ins = a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), a.get('GRBM_GUI_ACTIVE', 0.0), a.get('SQ_INSTS_MFMA', 0.0)
    if wc:
        print(f"   MFMA busy, per wave                  = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES)       = {100 * mf / (4 * wc):5.1f} %")
    if gui:
        print(f"   MFMA pipe utilisation (chip)         = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8) = {100 * mf / (1024 * gui / 8):5.1f} %")
        print(f"   launch duration                      = GRBM_GUI_ACTIVE / 8                                  = {gui / 8:9.0f} shader cycles")
    if ins:
        print(f"   cycles per MFMA instruction          = SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA             = {mf / ins:5.1f}")
        print(f"   VALU instructions per MFMA           = SQ_INSTS_VALU / SQ_INSTS_MFMA                        = {a.get('SQ_INSTS_VALU', 0.0) / ins:5.2f}")
PY
cat "$OUT/$NAME"
