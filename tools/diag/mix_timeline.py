#!/usr/bin/env python3
"""Gap analysis of the concurrent kernel mix from a rocprofv3 kernel trace of `bench.py --in-flight-only`:
  - share of the steady window with a WIDE kernel (bucket sweep or width-w NAF fold: the two that fill every CU on their own) in flight, and how many;
  - what runs in the remaining time: time-weighted census of the kernels resident while no wide kernel is;
  - for every sweep, what it waited for: the gap between the END of the previous kernel of its own stream and its START, split into the part during
    which another stream's wide kernel was running (the device was busy: queueing behind useful work) and the part with no wide kernel anywhere
    (a dependency / host round trip / launch latency: the device had room);
  - kernel-start delay under load: for the short kernels, duration in the mix against duration alone is NOT used (it needs a second trace); instead the
    census is reported per group with its average concurrency.
usage: mix_timeline.py kernel_trace.csv [window_s=2.0]  -> text on stdout"""
import collections, csv, sys
rows = []
with open(sys.argv[1]) as f:
    for x in csv.DictReader(f):
        name = x["Kernel_Name"].split("(")[0].replace("bpg::", "").replace("void ", "")
        stream = x.get("Thread_Id") or x.get("Stream_Id") or "0"     # one proving thread per HIP stream: the launching thread identifies the stream
        rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), name, stream))
rows.sort()
win = float(sys.argv[2]) * 1e9 if len(sys.argv) > 2 else 2.0e9
tend = max(e for s, e, n, q in rows if n.startswith("k_tt_finish"))
t0 = max(tend - win, min(s for s, e, n, q in rows if n.startswith("k_bucket_chunks")))
wide = lambda n: n.startswith("k_bucket_chunks") or "wnaf" in n


def group(n):
    if n.startswith("k_bucket_chunks"): return "sweep"
    if "wnaf" in n: return "first_fold"
    if n.startswith("k_fold"): return "folds"
    if n.startswith("k_bucket") or n.startswith("k_window"): return "epilogue"
    if n.startswith("k_msm") or n.startswith("k_scan"): return "sort"
    if n.startswith("k_tt"): return "tail"
    if n.startswith("__amd"): return "copies/fills"
    return "scalars/other"


sel = [(max(s, t0), min(e, tend), n, q) for s, e, n, q in rows if e > t0 and s < tend]
# sweep of events: at every instant the number of wide kernels and the multiset of narrow ones
ev = []
for k, (s, e, n, q) in enumerate(sel):
    ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort()
nwide, live = 0, set()
t_wide = collections.Counter(); census = collections.Counter(); conc_nowide = 0.0; any_kernel = 0.0
last = t0
for t, d, k in ev:
    dt = t - last
    if dt > 0:
        t_wide[min(nwide, 3)] += dt
        if live:
            any_kernel += dt
        if nwide == 0:
            conc_nowide += dt * len(live)
            for j in live:
                census[group(sel[j][2])] += dt
        last = t
    if d > 0:
        live.add(k); nwide += 1 if wide(sel[k][2]) else 0
    else:
        live.discard(k); nwide -= 1 if wide(sel[k][2]) else 0
span = tend - t0
proofs = sum(1 for s, e, n, q in sel if n.startswith("k_tt_finish")) / 12.0
print("steady window %.3f s, about %.0f proofs (%.2f ms per proof)" % (span / 1e9, proofs, span / 1e6 / max(proofs, 1)))
print("some kernel running: %.1f %% of the window" % (100 * any_kernel / span))
print("wide kernels (bucket sweep / width-w NAF fold) in flight:  0: %.1f %%   1: %.1f %%   2: %.1f %%   3+: %.1f %%" % tuple(100 * t_wide[k] / span for k in range(4)))
nw = t_wide[0]
print("while NO wide kernel is in flight (%.1f %% of the window): %.2f kernels resident on average; share of that time with a kernel of each group resident:" % (100 * nw / span, conc_nowide / max(nw, 1)))
for g, v in census.most_common():
    print("    %-14s %5.1f %%" % (g, 100 * v / max(nw, 1)))
# per stream: what each sweep waited for
by_stream = collections.defaultdict(list)
for s, e, n, q in rows:
    by_stream[q].append((s, e, n))
wide_iv = sorted((s, e) for s, e, n, q in rows if wide(n))


def covered(a, b):          # length of [a, b) covered by wide kernels (of any stream)
    tot, cur = 0, a
    for s, e in wide_iv:
        if e <= cur:
            continue
        if s >= b:
            break
        lo, hi = max(s, cur), min(e, b)
        if hi > lo:
            tot += hi - lo; cur = hi
    return tot


gaps = []
for q, ks in by_stream.items():
    ks.sort()
    for i in range(1, len(ks)):
        if ks[i][2].startswith("k_bucket_chunks") and t0 <= ks[i][0] < tend:
            a, b = ks[i - 1][1], ks[i][0]
            if b > a:
                gaps.append((b - a, covered(a, b), ks[i - 1][2]))
            else:
                gaps.append((0, 0, ks[i - 1][2]))              # back to back (the profiler's end stamp of one dispatch may even lie after the next one's start)
if gaps:
    tot = sum(g[0] for g in gaps); cov = sum(g[1] for g in gaps); waited = [g for g in gaps if g[0] > 0]
    print("sweeps in the window: %d; %d of them (%.1f %%) start back to back with the previous kernel of their stream; the others wait %.1f us on average; all gaps together: %.3f ms per proof"
          % (len(gaps), len(gaps) - len(waited), 100.0 * (len(gaps) - len(waited)) / len(gaps), (tot / len(waited) / 1e3) if waited else 0.0, tot / 1e6 / max(proofs, 1)))
    if tot:
        print("    of the waiting time, another stream's wide kernel was running %.1f %% (device busy), none %.1f %% (dependency, host, launch)" % (100 * cov / tot, 100 * (tot - cov) / tot))
    prev = collections.Counter()
    for g in gaps:
        prev[g[2]] += 1
    print("    kernel that precedes a sweep on its stream: " + ", ".join("%s x%d" % kv for kv in prev.most_common(4)))
# per stream: share of the window with no kernel of that stream resident (host-side work between rounds: encode L/R, transcript, challenge, recoding)
idle = []
for q, ks in by_stream.items():
    ks = sorted((max(s, t0), min(e, tend)) for s, e, n in ks if e > t0 and s < tend)
    if len(ks) < 100:
        continue
    busy, cur = 0, t0
    for s_, e_ in ks:
        if e_ > cur:
            busy += e_ - max(s_, cur); cur = e_
    idle.append(100.0 * (span - busy) / span)
if idle:
    print("proving streams: %d; share of the window a stream has no kernel resident (its host thread works between rounds): %s %%" % (len(idle), ", ".join("%.1f" % x for x in sorted(idle))))
# how long do kernels of each group take in the mix (duration = residency, not work)
dur = collections.defaultdict(list)
for s, e, n, q in sel:
    dur[n].append(e - s)
print("residency per launch in the mix (us): " + ", ".join("%s %.0f" % (n, sum(v) / len(v) / 1e3) for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]))
