"""Timeline of the headline burst from a rocprofv3 kernel trace (profiles: tools/profile_round.sh stats -> gpurun_out/<tag>_stats/<tag>_kernel_trace.csv):
per 20 ms bin of the last 0.64 s before the last tail kernel, the share of time with NO kernel running and the average number of kernels of each
group running.  usage: burst_timeline.py kernel_trace.csv"""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for x in csv.DictReader(f):
        rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"].split("(")[0].replace("bpg::", "").replace("void ", "")))
rows.sort()
tend = max(e for s, e, n in rows if "k_tt_finish" in n)
win, B = 0.64e9, 20e6
t0 = tend - win
nb = int(win / B)


def group(n):
    if "k_bucket_chunks" in n: return "sweep"
    if "wnaf" in n: return "first_fold"
    if "k_fold" in n: return "folds"
    if "k_bucket" in n or "k_window" in n: return "epilogue"
    if "k_msm" in n or "k_scan" in n: return "sort"
    if "k_tt" in n: return "tail"
    return "other"


bins = [collections.Counter() for _ in range(nb)]
edges = []
for s, e, n in rows:
    if e <= t0 or s >= tend:
        continue
    s = max(s, t0); e = min(e, tend)
    for b in range(int((s - t0) / B), min(int((e - t0) / B), nb - 1) + 1):
        lo, hi = max(s, t0 + b * B), min(e, t0 + (b + 1) * B)
        if hi > lo:
            bins[b][group(n)] += hi - lo
    edges += [(s, 1), (e, -1)]
edges.sort()
idle = [0.0] * nb
depth, last = 0, t0
for t, d in edges:
    if depth == 0 and t > last:
        for b in range(int((last - t0) / B), min(int((t - t0) / B), nb - 1) + 1):
            lo, hi = max(last, t0 + b * B), min(t, t0 + (b + 1) * B)
            if hi > lo:
                idle[b] += hi - lo
    depth += d
    if depth == 0:
        last = t
names = ["sweep", "first_fold", "folds", "epilogue", "sort", "tail", "other"]
print("  ms   idle   " + "  ".join("%10s" % n for n in names))
for b in range(nb):
    print("%4d  %5.1f%%  " % (b * B / 1e6, idle[b] / B * 100) + "  ".join("%10.2f" % (bins[b][n] / B) for n in names))
