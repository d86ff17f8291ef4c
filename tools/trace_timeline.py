"""Coarse timeline of a rocprofv3 kernel trace (csv): per time bin, how much kernel time of each kind ran
(sum of durations overlapping the bin / bin length: > 1 means kernels ran side by side).
usage: trace_timeline.py <kernel_trace.csv> [bin_ms] [last_ms]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
bin_ns = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 10e6
last_ns = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else None
def kind(n):
    for k, v in (("accumulate", "acc"), ("lm_step", "lm"), ("packet_jobs_kernel<4", "knn4"), ("packet_jobs_kernel<20", "knn20"),
                 ("packet_jobs_kernel<1", "knn1"), ("em_weight", "wgt"), ("proj_jobs", "proj"), ("cov_jobs", "cov"), ("copyBuffer", "copy"),
                 ("count_active", "cnt")):
        if k in n: return v
    return "other"
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind(r["Kernel_Name"])) for r in rows]
t1 = max(e for _, e, _ in ev)
t0 = t1 - last_ns if last_ns else min(s for s, _, _ in ev)
nb = int((t1 - t0) / bin_ns) + 1
kinds = ["knn20", "cov", "proj", "knn4", "wgt", "acc", "lm", "cnt", "copy", "other"]
acc = [collections.Counter() for _ in range(nb)]
for s, e, k in ev:
    if e <= t0: continue
    s = max(s, t0)
    b = int((s - t0) / bin_ns)
    while s < e and b < nb:
        be = t0 + (b + 1) * bin_ns
        acc[b][k] += min(e, be) - s
        s = be; b += 1
print("bin_ms  " + " ".join(f"{k:>6s}" for k in kinds) + "   total")
for b in range(nb):
    print(f"{b * bin_ns / 1e6:6.0f}  " + " ".join(f"{acc[b][k] / bin_ns:6.2f}" for k in kinds) + f"  {sum(acc[b].values()) / bin_ns:6.2f}")
# optional 4th/5th argument: zoom window [from_ms, to_ms) relative to the start of the analysed span, fine bins
if len(sys.argv) > 5:
    z0, z1 = t0 + float(sys.argv[4]) * 1e6, t0 + float(sys.argv[5]) * 1e6
    fb = float(sys.argv[6]) * 1e6 if len(sys.argv) > 6 else 0.25e6
    nz = int((z1 - z0) / fb)
    za = [collections.Counter() for _ in range(nz)]
    for s, e, k in ev:
        if e <= z0 or s >= z1: continue
        s = max(s, z0)
        b = int((s - z0) / fb)
        while s < min(e, z1) and b < nz:
            be = z0 + (b + 1) * fb
            za[b][k] += min(e, be) - s
            s = be; b += 1
    print(f"zoom ({fb / 1e6} ms bins)")
    for b in range(nz):
        print(f"{(z0 - t0) / 1e6 + b * fb / 1e6:7.2f}  " + " ".join(f"{za[b][k] / fb:6.2f}" for k in kinds))
