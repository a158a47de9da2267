"""Where is the GPU idle?  From a rocprofv3 --kernel-trace csv: the gaps during which NO kernel runs on any stream,
grouped by the kernel that ends before and the kernel that starts after the gap.
usage: python tools/trace_gaps.py <dir or *_kernel_trace.csv> [--last-ms T] [--top N]"""
import collections, csv, glob, os, sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return (n[:n.rfind(">") + 1] if "<" in n and n.rfind(">") > 0 else n.split("(")[0])[:60]


def main():
    path = sys.argv[1]
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t1 = int(rows[-1]["End_Timestamp"])
    if "--last-ms" in sys.argv:
        cut = t1 - float(sys.argv[sys.argv.index("--last-ms") + 1]) * 1e6
        rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    cur_e, cur_name = int(rows[0]["End_Timestamp"]), short(rows[0]["Kernel_Name"])
    gaps = []
    for r in rows[1:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > cur_e:
            gaps.append((s - cur_e, cur_name, short(r["Kernel_Name"])))
        if e > cur_e:
            cur_e, cur_name = e, short(r["Kernel_Name"])
    tot = sum(g[0] for g in gaps)
    span = t1 - int(rows[0]["Start_Timestamp"])
    print(f"span {span / 1e6:.1f} ms, {len(rows)} kernels, {len(gaps)} all-idle gaps, {tot / 1e6:.2f} ms idle ({100 * tot / span:.1f} %)")
    for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 200), (200, 1e9)):
        sel = [g[0] for g in gaps if lo * 1e3 <= g[0] < hi * 1e3]
        print(f"  gaps {lo:>4}-{hi if hi < 1e9 else 'inf':>4} us: n={len(sel):5d}  sum {sum(sel) / 1e6:7.2f} ms")
    by_after, by_before = collections.defaultdict(lambda: [0, 0]), collections.defaultdict(lambda: [0, 0])
    for d, before, after in gaps:
        by_after[after][0] += d; by_after[after][1] += 1
        by_before[before][0] += d; by_before[before][1] += 1
    print("idle before the launch of (kernel that ends the gap):")
    for k, (d, n) in sorted(by_after.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"  {d / 1e6:7.2f} ms  n={n:5d}  avg {d / n / 1e3:6.1f} us  {k}")
    print("idle after the end of:")
    for k, (d, n) in sorted(by_before.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"  {d / 1e6:7.2f} ms  n={n:5d}  avg {d / n / 1e3:6.1f} us  {k}")
    print("largest gaps:")
    for d, before, after in sorted(gaps, reverse=True)[:15]:
        print(f"  {d / 1e3:8.1f} us  {before}  ->  {after}")


main()
