#!/usr/bin/env python3
"""Per-kernel VALU instruction budget of a proof from a rocprofv3 --pmc pass over tools/diag/insts_workload.py.
usage: insts_summarize.py <counter_collection.csv> <proofs> [<kernel_trace.csv>]
Prints, per kernel: launches per proof, VALU wave-instructions per proof (SQ_INSTS_VALU), share, and - with SQ_BUSY_CYCLES / SQ_WAVE_CYCLES /
SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE in the pass - issue utilisation and the clock the dispatch ran at."""
import collections, csv, json, re, sys
rows = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("bpg::", "")
    rows[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), name)
    if key not in seen:
        seen.add(key); launches[name] += 1
proofs = float(sys.argv[2])
dur = collections.defaultdict(float)
if len(sys.argv) > 3:
    for r in csv.DictReader(open(sys.argv[3])):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("bpg::", "")
        dur[name] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6
bench = rows.get("k_bench_fe_mul")
tot = sum(v.get("SQ_INSTS_VALU", 0) for k, v in rows.items() if k != "k_bench_fe_mul" and k.startswith("k_"))
out = {"proofs": proofs, "valu_insts_per_proof": tot / proofs, "kernels": {}}
print("%-26s %7s %14s %6s %9s %9s %8s" % ("kernel", "n/proof", "VALU/proof", "share", "ms/proof", "Ginst/s", "clk GHz"))
for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    n = launches[k]
    per = v.get("SQ_INSTS_VALU", 0) / (1 if k == "k_bench_fe_mul" else proofs)
    ms = dur.get(k, 0) / (1 if k == "k_bench_fe_mul" else proofs)
    clk = v.get("GRBM_GUI_ACTIVE", 0) / 8 / (dur[k] * 1e-3) / 1e9 if dur.get(k) else 0
    rate = per / (ms * 1e-3) / 1e9 if ms else 0
    out["kernels"][k] = {"launches_per_proof": n / proofs, "valu_insts": per, "ms": ms, "ginst_per_s": rate, "clock_ghz": clk,
                         "counters": {c: x / (1 if k == "k_bench_fe_mul" else proofs) for c, x in v.items()}}
    print("%-26s %7.1f %14.0f %5.1f%% %9.3f %9.1f %8.2f" % (k, n / proofs, per, 100 * per / tot if k != "k_bench_fe_mul" and tot else 0, ms, rate, clk))
if bench and dur.get("k_bench_fe_mul"):
    peak = bench["SQ_INSTS_VALU"] / (dur["k_bench_fe_mul"] * 1e-3)
    out["issue_peak_inst_per_s"] = peak
    out["floor_ms_per_proof_at_bench_rate"] = tot / proofs / peak * 1e3
    print("k_bench_fe_mul issues %.1f G VALU wave-instructions/s; a proof's %.3f G at that rate: %.2f ms" % (peak / 1e9, tot / proofs / 1e9, tot / proofs / peak * 1e3))
json.dump(out, open(sys.argv[1] + ".summary.json", "w"), indent=1)
