#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (two separate runs, as MI355X_MICROARCH.md prescribes) into
pmc_traffic.json: HBM-side bytes per launch for each kernel, stamped with the hash of the sources the passes were taken from.
  fetch bytes = FETCH_SIZE * 1024 * factor,  write bytes = WRITE_SIZE * 1024
FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-byte request, so for EVERY kernel
    hbm_bytes_per_launch = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024          (the guide's correction; bytes moved in 128-byte lines)
For kernels that GATHER rows at random (the bucket sweep, the table rounds, the width-w NAF fold) a second, separately named figure is kept:
    useful_fetch_bytes_per_launch = FETCH_SIZE x 1024 x gather96_factor
with the factor measured by tools/calib/gather_calib.hip on a known byte count in that access shape (<tag>_fetch_calibration.json: a 96-byte row
touches 1.5 lines, each tallied at 64 B).  It estimates the bytes the kernel asked for, NOT HBM traffic - the calibration table (201 MB) sits in
the Infinity Cache - and is never added into hbm_bytes.
usage: pmc_summarize.py <fetch_counter_collection.csv> <write_counter_collection.csv> <fetch_calibration.json|-> <out.json> [<bench --in-flight-only line.json>]
With the fifth argument (the JSON line of the profiled `bench.py --in-flight-only` run) the summary also says how many proofs the passes covered
(the timed ones, one warm-up per proving stream and the first proof of the process) and the HBM bytes per proof of the concurrent mix."""
import collections, csv, json, pathlib, re, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))

GATHER_KERNELS = ("k_bucket_chunks", "k_tt_round", "k_tt_round8", "k_tt_commit3", "k_fold_points_wnaf")


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
    raw = f.get(k, 0.0) * 1024
    fetch, write = raw * 2.0, w.get(k, 0.0) * 1024
    out[k] = {"launches": n, "fetch_size_raw_per_launch": raw / n, "fetch_bytes_per_launch": fetch / n, "write_bytes_per_launch": write / n,
              "hbm_bytes_per_launch": (fetch + write) / n, "total_hbm_bytes": fetch + write, "access": "gather" if k in GATHER_KERNELS else "stream"}
    if k in GATHER_KERNELS:
        out[k]["useful_fetch_bytes_per_launch"] = raw * gather_factor / n
if len(sys.argv) > 5:
    line = json.loads(open(sys.argv[5]).read().strip().splitlines()[-1])["in_flight"]
    proofs = line["proofs"] + line["proofs_in_flight"] + 1
    total = sum(v["total_hbm_bytes"] for k, v in out.items() if k != "_meta" and k.startswith("k_"))
    out["_meta"].update({"proofs_profiled": proofs, "hbm_bytes_per_proof": total / proofs, "proving_streams": line["proofs_in_flight"],
                         "mix": "bench.py --in-flight-only: %d proving streams, kernels of different proofs share the CUs" % line["proofs_in_flight"]})
json.dump(out, open(sys.argv[4], "w"), indent=1, sort_keys=True)
print(calib_note)
for k, v in sorted(((k, v) for k, v in out.items() if k != "_meta"), key=lambda kv: -kv[1]["total_hbm_bytes"])[:14]:
    print("%-28s launches %4d  fetch/launch %12.0f B (x2)  write/launch %12.0f B%s" % (k, v["launches"], v["fetch_bytes_per_launch"], v["write_bytes_per_launch"],
                                                                                         "  useful %12.0f B" % v["useful_fetch_bytes_per_launch"] if "useful_fetch_bytes_per_launch" in v else ""))
