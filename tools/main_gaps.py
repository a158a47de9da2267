"""Gaps of the busiest stream in a rocprofv3 kernel trace: for every gap above a threshold, the kernels that bracket it and what
the other streams ran meanwhile (their busy fraction of the gap and their longest kernel).
usage: python tools/main_gaps.py <kernel_trace.csv> [--last-ms 48] [--min-us 50] [--top 40]"""
import csv, sys, collections

def arg(name, default):
    return float(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t1 = int(rows[-1]["End_Timestamp"])
cut = t1 - arg("--last-ms", 48.0) * 1e6
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
key = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
tot = collections.defaultdict(float)
for r in rows:
    tot[r[key]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
main = max(tot, key=tot.get)
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
mk = [r for r in rows if r[key] == main]
ot = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r[key]) for r in rows if r[key] != main]
gaps = []
for a, b in zip(mk[:-1], mk[1:]):
    g0, g1 = int(a["End_Timestamp"]), int(b["Start_Timestamp"])
    if g1 - g0 >= arg("--min-us", 50.0) * 1e3:
        ev = sorted((max(s, g0), min(e, g1), n) for s, e, n, _ in ot if e > g0 and s < g1)
        busy, cur = 0, g0
        for s, e, _ in ev:
            if e > cur:
                busy += e - max(s, cur); cur = e
        longest = max(ev, key=lambda t: t[1] - t[0], default=(0, 0, "-"))
        gaps.append((g1 - g0, short(a["Kernel_Name"]), short(b["Kernel_Name"]), busy / (g1 - g0), len(ev), longest[2], (longest[1] - longest[0]) / 1e3, (g0 - cut) / 1e6))
print(f"main stream {main}: {len(mk)} kernels; {len(gaps)} gaps >= {arg('--min-us', 50.0):.0f} us, sum {sum(g[0] for g in gaps) / 1e6:.2f} ms; "
      f"of that with other streams busy: {sum(g[0] * g[3] for g in gaps) / 1e6:.2f} ms")
for g in sorted(gaps, key=lambda g: -g[0])[:int(arg("--top", 40))]:
    print(f"{g[0] / 1e3:8.1f} us at {g[7]:6.2f} ms  after {g[1]:48s} before {g[2]:48s} others busy {100 * g[3]:3.0f}% ({g[4]} kernels, longest {g[5]} {g[6]:.0f} us)")
