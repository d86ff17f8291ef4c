"""Analyse a rocprofv3 kernel trace (csv): per-queue gaps and device-wide concurrency.
usage: trace_concurrency.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
byq = collections.defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    short = "acc" if "accumulate" in name else "lm" if "lm_step" in name else "knn4" if "packet_kernel<4" in name else \
        "knn20" if "packet_kernel<20" in name else "other"
    byq[r["Queue_Id"]].append((s, e, short))
    ev.append((s, e, short))
t0 = min(s for s, _, _ in ev); t1 = max(e for _, e, _ in ev)
print(f"kernels {len(ev)}  queues {len(byq)}  span {(t1 - t0) / 1e6:.2f} ms  sum of durations {sum(e - s for s, e, _ in ev) / 1e6:.2f} ms")
# concurrency histogram (time-weighted)
pts = sorted([(s, 1) for s, _, _ in ev] + [(e, -1) for _, e, _ in ev])
cur = 0; last = pts[0][0]; hist = collections.Counter()
for t, d in pts:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print("time share by number of kernels running:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
# gaps inside each queue between consecutive kernels, by (prev -> next) type
gaps = collections.defaultdict(list)
for q, lst in byq.items():
    lst.sort()
    for (s0, e0, n0), (s1, e1, n1) in zip(lst, lst[1:]):
        gaps[(n0, n1)].append(s1 - e0)
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{k[0]:>6s} -> {k[1]:<6s} n={len(v):5d} median gap {v[len(v) // 2] / 1e3:8.2f} us  mean {sum(v) / len(v) / 1e3:8.2f} us  total {sum(v) / 1e6:8.2f} ms")
dur = collections.defaultdict(list)
for s, e, n in ev:
    dur[n].append(e - s)
for n, v in dur.items():
    v.sort()
    print(f"{n:>6s} n={len(v):5d} median {v[len(v) // 2] / 1e3:8.2f} us mean {sum(v) / len(v) / 1e3:8.2f} us")
