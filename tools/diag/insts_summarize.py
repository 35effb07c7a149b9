#!/usr/bin/env python3
"""Per-kernel VALU instruction budget of ONE PROOF from a rocprofv3 --pmc pass over tools/diag/insts_workload.py.
usage: insts_summarize.py <counter_collection.csv> <proofs> [<kernel_trace.csv>]
One-off work of the process is left out: the kernels that build generator tables (k_gens_derive, k_odd_*, k_dbl_times, k_init_bases, the 2^21-point
k_normalize_niels launches of the table build - a proof's own three normalisations are at most 2^18 points), the circuit upload (k_csc_*, k_sc_from_bytes)
and the one verification the workload ends with (k_decompress, k_ipa_s, k_verify_scalars, k_flatten_const).  Prints, per kernel: launches per proof,
VALU wave-instructions per proof (SQ_INSTS_VALU), share, time under the profiler (kernels run one at a time there), issue rate and the clock
the dispatches ran at (GRBM_GUI_ACTIVE / 8 / duration); then the proof's total against the issue rate of the field-multiplication microbenchmark."""
import collections, csv, json, re, sys
SETUP = ("k_gens_derive", "k_odd_start", "k_odd_start_ext", "k_odd_step", "k_dbl_times", "k_init_bases", "k_csc_count", "k_csc_fill", "k_csc_colptr", "k_sc_from_bytes",
         "k_decompress", "k_ipa_s", "k_verify_scalars", "k_flatten_const", "k_tt_bases8", "k_tt_multiples8", "k_compress_niels")
clean = lambda n: re.sub(r"\(.*", "", n).replace("void ", "").replace("bpg::", "")
disp = collections.defaultdict(dict); dname = {}
for r in csv.DictReader(open(sys.argv[1])):
    d = r["Dispatch_Id"]; dname[d] = clean(r["Kernel_Name"])
    disp[d][r["Counter_Name"]] = disp[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
proofs = float(sys.argv[2])
ddur = {}
if len(sys.argv) > 3:
    for r in csv.DictReader(open(sys.argv[3])):
        ddur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6
biggest = collections.defaultdict(float)
for d, c in disp.items():
    biggest[dname[d]] = max(biggest[dname[d]], c.get("SQ_INSTS_VALU", 0.0))
rows = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.Counter(); dur = collections.defaultdict(float); skipped = collections.Counter()
for d, c in disp.items():
    n = dname[d]
    if n in SETUP or n.startswith("__amd") or (n == "k_normalize_niels" and c.get("SQ_INSTS_VALU", 0.0) > 0.25 * biggest[n] and biggest[n] > 1e8):
        skipped[n] += 1
        continue
    for k, v in c.items():
        rows[n][k] += v
    launches[n] += 1; dur[n] += ddur.get(d, 0.0)
bench = rows.pop("k_bench_fe_mul", None); bdur = dur.pop("k_bench_fe_mul", 0.0); launches.pop("k_bench_fe_mul", None)
tot = sum(v.get("SQ_INSTS_VALU", 0) for v in rows.values())
out = {"proofs": proofs, "valu_wave_instructions_per_proof": tot / proofs, "kernels": {}, "left_out": dict(skipped)}
print("%-28s %7s %14s %6s %9s %9s %8s" % ("kernel", "n/proof", "VALU/proof", "share", "ms/proof", "Ginst/s", "clk GHz"))
for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    per = v.get("SQ_INSTS_VALU", 0) / proofs; ms = dur[k] / proofs
    clk = v.get("GRBM_GUI_ACTIVE", 0) / 8 / (dur[k] * 1e-3) / 1e9 if dur[k] else 0
    rate = per / (ms * 1e-3) / 1e9 if ms else 0
    out["kernels"][k] = {"launches_per_proof": launches[k] / proofs, "valu_wave_instructions": per, "ms_under_profiler": ms, "ginst_per_s": rate, "clock_ghz": clk}
    print("%-28s %7.1f %14.0f %5.1f%% %9.3f %9.1f %8.2f" % (k, launches[k] / proofs, per, 100 * per * proofs / tot if tot else 0, ms, rate, clk))
print("one proof: %.3f G VALU wave-instructions in %d launches (left out as one-off work: %s)" % (tot / proofs / 1e9, sum(launches.values()) / proofs, ", ".join("%s x%d" % kv for kv in sorted(skipped.items()))))
if bench and bdur:
    peak = bench["SQ_INSTS_VALU"] / (bdur * 1e-3); bclk = bench.get("GRBM_GUI_ACTIVE", 0) / 8 / (bdur * 1e-3) / 1e9
    out["issue_rate_fe_mul_microbenchmark"] = peak; out["floor_ms_per_proof_at_that_rate"] = tot / proofs / peak * 1e3
    print("k_bench_fe_mul (dependent-free field multiplications, the yardstick of integer-VALU issue) issues %.1f G VALU wave-instructions/s at %.2f GHz;" % (peak / 1e9, bclk))
    print("a proof's %.3f G at that rate: %.2f ms" % (tot / proofs / 1e9, tot / proofs / peak * 1e3))
json.dump(out, open(sys.argv[1] + ".summary.json", "w"), indent=1)
