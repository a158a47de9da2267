"""Per-stream activity over time from a rocprofv3 --kernel-trace csv: kernel-busy milliseconds of every stream in consecutive time buckets of the
LAST step of the trace (who runs when: main / aux sweep / weight-gradient / peer streams).
usage: python tools/stream_timeline.py <dir or *_kernel_trace.csv> --last-ms <step ms> [--bucket-ms 2]"""
import csv, glob, os, sys, collections

path = sys.argv[1]
last = float(sys.argv[sys.argv.index("--last-ms") + 1])
bucket = float(sys.argv[sys.argv.index("--bucket-ms") + 1]) if "--bucket-ms" in sys.argv else 2.0
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
key = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
t1 = max(int(r["End_Timestamp"]) for r in rows)
t0 = t1 - last * 1e6
rows = [r for r in rows if int(r["End_Timestamp"]) > t0]
nb = int(last / bucket) + 1
busy = collections.defaultdict(lambda: [0.0] * nb)
count = collections.defaultdict(lambda: [0] * nb)
for r in rows:
    s, e = max(int(r["Start_Timestamp"]), t0), int(r["End_Timestamp"])
    b = int((s - t0) / 1e6 / bucket)
    count[r[key]][min(b, nb - 1)] += 1
    while s < e:
        b = int((s - t0) / 1e6 / bucket)
        be = t0 + (b + 1) * bucket * 1e6
        busy[r[key]][min(b, nb - 1)] += (min(e, be) - s) / 1e6
        s = min(e, be)
streams = sorted(busy, key=lambda k: -sum(busy[k]))
print("bucket start (ms) | kernel-busy ms (launches) per stream: " + "  ".join(f"s{k}" for k in streams))
for b in range(nb):
    print(f"{b * bucket:6.1f} | " + "  ".join(f"{busy[k][b]:5.2f} ({count[k][b]:3d})" for k in streams))
print("total            | " + "  ".join(f"{sum(busy[k]):5.1f} ({sum(count[k]):4d})" for k in streams))
