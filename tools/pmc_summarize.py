#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (two separate runs, as MI355X_MICROARCH.md prescribes) into
pmc_traffic.json: HBM-side bytes per launch for each kernel, stamped with the hash of the sources the passes were taken from.
  fetch bytes = FETCH_SIZE * 1024 * factor,  write bytes = WRITE_SIZE * 1024
FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-byte request: factor 2 for wide coalesced reads
(guide, section HBM).  For kernels that GATHER 96-byte points at random (the bucket sweep, the table rounds) the factor is the one
measured by tools/calib/gather_calib.hip on a known byte count in that access shape (<tag>_fetch_calibration.json); the value stored
is then an estimate of USEFUL bytes delivered; `fetch_bytes_x2` keeps the guide's doubling next to it (bytes moved in 128-byte lines).
usage: pmc_summarize.py <fetch_counter_collection.csv> <write_counter_collection.csv> <fetch_calibration.json|-> <out.json>"""
import collections, csv, json, pathlib, re, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))

GATHER_KERNELS = ("k_bucket_chunks", "k_tt_round", "k_tt_commit3")


def load(path, counter):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("bpg::", "")
        tot[name] += float(r["Counter_Value"]); cnt[name] += 1
    return tot, cnt


f, fc = load(sys.argv[1], "FETCH_SIZE")
w, wc = load(sys.argv[2], "WRITE_SIZE")
stream_factor, gather_factor, calib_note = 2.0, 2.0, "no calibration file: the guide's x2 everywhere"
if sys.argv[3] != "-" and pathlib.Path(sys.argv[3]).exists():
    c = json.loads(open(sys.argv[3]).read())["summary"]
    if c.get("stream_factor"):
        stream_factor = c["stream_factor"]
    if c.get("gather96_factor"):
        gather_factor = c["gather96_factor"]
    calib_note = "FETCH_SIZE factors measured by tools/calib/gather_calib.hip: %.3f streaming, %.3f for random 96-byte rows" % (stream_factor, gather_factor)
from bench import source_hash
out = {"_meta": {"source_hash": source_hash(), "note": calib_note, "stream_factor": stream_factor, "gather96_factor": gather_factor,
                 "gather_kernels": list(GATHER_KERNELS)}}
for k in sorted(set(f) | set(w)):
    n = max(fc.get(k, 0), wc.get(k, 0), 1)
    fac = gather_factor if k in GATHER_KERNELS else stream_factor
    raw = f.get(k, 0.0) * 1024
    fetch, write = raw * fac, w.get(k, 0.0) * 1024
    out[k] = {"launches": n, "fetch_size_raw_per_launch": raw / n, "fetch_factor": fac, "fetch_bytes_per_launch": fetch / n, "fetch_bytes_x2_per_launch": raw * 2 / n,
              "write_bytes_per_launch": write / n, "hbm_bytes_per_launch": (fetch + write) / n, "total_hbm_bytes": fetch + write}
json.dump(out, open(sys.argv[4], "w"), indent=1, sort_keys=True)
print(calib_note)
for k, v in sorted(((k, v) for k, v in out.items() if k != "_meta"), key=lambda kv: -kv[1]["total_hbm_bytes"])[:14]:
    print("%-28s launches %4d  fetch/launch %12.0f B (x%.2f)  write/launch %12.0f B" % (k, v["launches"], v["fetch_bytes_per_launch"], v["fetch_factor"], v["write_bytes_per_launch"]))
