#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (two separate runs, as MI355X_MICROARCH.md prescribes)
into profiles/pmc_traffic.json: HBM bytes per launch for each kernel.
  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, hence the factor 2
(guide section HBM; calibrated there for wide coalesced 16-B-per-lane reads - our gathers of 96-B points are 16-B loads
but not lane-contiguous, so treat the absolute value as +-2x and compare it with the algorithmic bytes only for order of magnitude).
usage: pmc_summarize.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv, json, sys, collections, re

def load(path, counter):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("bpg::", "")
        tot[name] += float(r["Counter_Value"]); cnt[name] += 1
    return tot, cnt

f, fc = load(sys.argv[1], "FETCH_SIZE")
w, wc = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    n = max(fc.get(k, 0), wc.get(k, 0), 1)
    fetch, write = f.get(k, 0.0) * 1024 * 2, w.get(k, 0.0) * 1024
    out[k] = {"launches": n, "fetch_bytes_per_launch": fetch / n, "write_bytes_per_launch": write / n,
              "hbm_bytes_per_launch": (fetch + write) / n, "total_hbm_bytes": fetch + write}
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["total_hbm_bytes"])[:12]:
    print("%-28s launches %4d  HBM/launch %12.0f B  total %14.0f B" % (k, v["launches"], v["hbm_bytes_per_launch"], v["total_hbm_bytes"]))
