"""Aggregate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in separate runs) into per-kernel HBM bytes per launch.

  python tools/pmc_summary.py <dir with fetch pass> <dir with write pass> <out json>

Per MI355X_MICROARCH.md (HBM / rocprofv3): the counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a
wide coalesced streaming read, so the read side is doubled ("fetch_x2"); WRITE_SIZE is taken as is (uncalibrated)."""
import csv, glob, json, os, re, sys, collections


def load(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for fn in files:
        for r in csv.DictReader(open(fn)):
            if r.get("Counter_Name") != counter:
                continue
            n = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "")
            n = re.sub(r"\(.*$", "", n)
            a = agg[n]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


def main():
    fdir, wdir, out = sys.argv[1:4]
    f, w = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    ks = {}
    for n in sorted(set(f) | set(w)):
        fk = f[n][1] / max(f[n][0], 1) if n in f else 0.0
        wk = w[n][1] / max(w[n][0], 1) if n in w else 0.0
        ks[n] = {"launches_sampled": int(max(f.get(n, [0])[0], w.get(n, [0])[0])), "fetch_kib_raw_per_launch": round(fk, 2),
                 "write_kib_per_launch": round(wk, 2), "hbm_bytes_per_launch": round((2.0 * fk + wk) * 1024.0)}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH x2 correction)",
               "kernels": ks}, open(out, "w"), indent=1)
    top = sorted(ks.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_sampled"])[:12]
    for n, v in top:
        print(f"{n[:70]:70s} n={v['launches_sampled']:5d} bytes/launch={v['hbm_bytes_per_launch']/1e6:9.2f} MB")


if __name__ == "__main__":
    main()
