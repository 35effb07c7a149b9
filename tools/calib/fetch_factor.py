#!/usr/bin/env python3
"""FETCH_SIZE calibration: known bytes of tools/calib/gather_calib (its JSON line) against rocprofv3 --pmc FETCH_SIZE of the same run.
usage: fetch_factor.py <gather_calib stdout json> <counter_collection.csv> <out.json>
factor[k] = known bytes / (FETCH_SIZE * 1024): what FETCH_SIZE has to be multiplied with, for that access shape, to give bytes."""
import collections, csv, json, re, sys
known = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
tot, cnt = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(sys.argv[2])):
    if r["Counter_Name"] != "FETCH_SIZE":
        continue
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    tot[name] += float(r["Counter_Value"]); cnt[name] += 1
out = {"table_bytes": known["table_bytes"], "rows_gathered": known["rows_gathered"], "ms": known["ms"], "GBps": known["GBps"], "kernels": {}}
for k, kb in known["known_bytes"].items():
    if cnt.get(k):
        fetch = tot[k] / cnt[k] * 1024
        out["kernels"][k] = {"known_bytes": kb, "FETCH_SIZE_bytes": fetch, "factor": kb / fetch, "launches": cnt[k]}
g = out["kernels"].get("k_calib_gather<6>")
s = out["kernels"].get("k_calib_stream")
out["summary"] = {"stream_factor": s and s["factor"], "gather96_factor": g and g["factor"],
                  "note": "factor = known bytes / FETCH_SIZE bytes. The guide's x2 is the stream factor; for 96-byte rows gathered at random from a 201 MB table "
                          "(k_bucket_chunks' access shape) FETCH_SIZE has to be multiplied with gather96_factor to give the 96 useful bytes per row; "
                          "128-byte lines touched per 96-byte row: 1.5 on average (rows at 96 i), i.e. 192 B moved per row if every line is fetched once"}
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
print(json.dumps(out["summary"]))
