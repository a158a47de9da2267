"""Summarise a rocprofv3 --kernel-trace csv: per-kernel totals per step, per-stream busy time, idle gaps.
usage: python tools/trace_summary.py <dir or *_kernel_trace.csv> [--steps K] [--skip-frac F] [--main-stream] [--forward]
--main-stream: the breakdown of the BUSIEST stream only (the critical path of the step: the other streams overlap it), with the
gaps between its consecutive kernels."""
import csv, glob, os, sys, collections

def main():
    path = sys.argv[1]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 1
    skip = float(sys.argv[sys.argv.index("--skip-frac") + 1]) if "--skip-frac" in sys.argv else 0.0
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
    cut = t0 + skip * (t1 - t0)
    if "--last-ms" in sys.argv:
        cut = t1 - float(sys.argv[sys.argv.index("--last-ms") + 1]) * 1e6
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
    if "--forward" in sys.argv:
        # the forward passes only: from the end of an optimizer launch to the first backward kernel of the losses (one stream, every kernel on the step's critical path)
        keep, inside, nwin = [], False, 0
        for r in rows:
            nm = r["Kernel_Name"]
            if "adamw_kernel" in nm:
                inside = True; nwin += 1
                continue
            if inside and ("seg_losses_bwd" in nm or "_bwd_kernel" in nm and "loss" in nm):
                inside = False
            if inside:
                keep.append(r)
        rows = keep
        steps = max(1, nwin - (1 if inside else 0))
        t1 = int(rows[-1]["End_Timestamp"])
        print(f"forward windows: {steps} (adamw_kernel .. first loss-backward kernel)")
    t0 = int(rows[0]["Start_Timestamp"])
    span = (t1 - t0) / 1e6
    if "--stream-rank" in sys.argv:
        # the breakdown of ONE stream, chosen by its rank in kernel time (0 = busiest = the main stream, 1 / 2 = the weight-gradient and aux-sweep streams)
        key = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
        tot_s = collections.defaultdict(float)
        for r in rows:
            tot_s[r[key]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        order = sorted(tot_s, key=tot_s.get, reverse=True)
        sid = order[min(int(sys.argv[sys.argv.index("--stream-rank") + 1]), len(order) - 1)]
        rows = [r for r in rows if r[key] == sid]
        print(f"stream {sid} (rank {order.index(sid)} by kernel time): {len(rows)} kernels, first at {(int(rows[0]['Start_Timestamp']) - t0) / 1e6:.2f} ms, last ends at "
              f"{(int(rows[-1]['End_Timestamp']) - t0) / 1e6:.2f} ms of the window")
    if "--main-stream" in sys.argv:
        key = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
        tot_s = collections.defaultdict(float)
        for r in rows:
            tot_s[r[key]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        main_id = max(tot_s, key=tot_s.get)
        rows = [r for r in rows if r[key] == main_id]
        g = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows[:-1], rows[1:])]
        g = [x for x in g if x > 0]
        small = [x for x in g if x < 20000]
        print(f"main stream {main_id}: {len(rows)} kernels, {len(rows) / steps:.0f}/step; gaps between consecutive kernels: {sum(g) / 1e6 / steps:.2f} ms/step "
              f"(of which < 20 us: n={len(small) / steps:.0f}/step, {sum(small) / 1e6 / steps:.2f} ms/step, median {sorted(small)[len(small) // 2] / 1e3:.1f} us)")
    per = collections.defaultdict(lambda: [0.0, 0])
    streams = collections.defaultdict(float)
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0] if "<" not in name else name[:name.rfind(">") + 1] if name.rfind(">") > 0 else name
        per[name][0] += d; per[name][1] += 1
        streams[r.get("Stream_Id", r.get("Queue_Id", "?"))] += d
    # union busy time
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
    busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    gaps = []
    for s, e in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"span {span:.1f} ms  busy(union) {busy/1e6:.1f} ms  idle {span-busy/1e6:.1f} ms  kernels {len(rows)}  steps {steps}")
    print("per-stream kernel time (ms):", {k: round(v, 1) for k, v in streams.items()})
    if gaps:
        gaps.sort()
        print(f"gaps: n={len(gaps)} median {gaps[len(gaps)//2]/1e3:.1f} us  p90 {gaps[int(len(gaps)*0.9)]/1e3:.1f} us  sum {sum(gaps)/1e6:.1f} ms")
    tot = sum(v[0] for v in per.values())
    print(f"sum of kernel durations {tot:.1f} ms  ({tot/steps:.1f} ms/step)")
    for name, (d, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 60]:
        print(f"{d/steps:8.2f} ms/step {n//steps:6d}/step {d/n*1e3:8.1f} us  {100*d/tot:5.1f}%  {name[:110]}")

main()
