#!/usr/bin/env python3
"""Copy a round's GPU evidence from gpurun_out/ (what tools/profile_round.sh and tools/diag/lone_proof.sh wrote on the GPU box) into profiles/ under the names
DESIGN.md cites, and refuse evidence that was not taken from the tree as it is now: every JSON that carries a source hash must carry bench.source_hash().
usage (build container, repo root):  python tools/collect_round.py r05        then  bash tools/sanitize_round.sh r05
The text files keep their first (descriptive) line from the copy already in profiles/, with the hash in it replaced."""
import json, pathlib, re, shutil, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench                                                   # noqa: E402
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
G, P = ROOT / "gpurun_out", ROOT / "profiles"
H = bench.source_hash()


def last_json(path):
    return json.loads(path.read_text().strip().splitlines()[-1])


bad = []
for name in ("%s_pmc_traffic.json" % tag, "%s_pmc_traffic_throughput.json" % tag):
    h = json.loads((G / name).read_text()).get("_meta", {}).get("source_hash")
    if h != H:
        bad.append("%s: sources %s, the tree is %s" % (name, h, H))
line = last_json(G / ("%s_bench_plain.json" % tag))
if line.get("source_hash") != H:
    bad.append("%s_bench_plain.json: sources %s, the tree is %s" % (tag, line.get("source_hash"), H))
if bad:
    sys.exit("evidence of other sources, re-take it:\n  " + "\n  ".join(bad))

plain = {"%s_bench_plain.json": "%s_bench_plain.json", "%s_bench_under_rocprof.json": "%s_bench_under_rocprof.json",
         "%s_bench_single_under_rocprof.json": "%s_bench_single_under_rocprof.json", "%s_bench_inflight_under_rocprof.json": "%s_bench_inflight_under_rocprof.json",
         "%s_pmc_traffic.json": "%s_pmc_traffic.json", "%s_pmc_traffic_throughput.json": "%s_pmc_traffic_throughput.json",
         "%s_fetch_calibration.json": "%s_fetch_calibration.json", "%s_inversion_cost.json": "%s_inversion_cost.json",
         "%s_insts_summary.json": "%s_instruction_budget.json", "%s_appetite.txt": "%s_appetite.txt", "%s_clock.txt": "%s_clock_and_power.jsonl",
         "%s_stats/%s_kernel_stats.csv": "%s_rocprof_kernel_stats.csv", "%s_stats_single/%s_kernel_stats.csv": "%s_rocprof_kernel_stats_single_stream.csv",
         "%s_stats_inflight/%s_kernel_stats.csv": "%s_rocprof_kernel_stats_throughput.csv"}
for src, dst in plain.items():
    s, d = G / (src.replace("%s", tag)), P / (dst % tag)
    if s.exists():
        shutil.copy(s, d)
    else:
        print("missing:", s)
# bench.py quotes the counter traffic from the un-prefixed names (and only when their source hash is the tree's)
shutil.copy(P / ("%s_pmc_traffic.json" % tag), P / "pmc_traffic.json")
shutil.copy(P / ("%s_pmc_traffic_throughput.json" % tag), P / "pmc_traffic_throughput.json")


def with_header(dst, body_path, header_lines):
    old = (P / dst).read_text().split("\n") if (P / dst).exists() else []
    hdr = [re.sub(r"bench\.source_hash \w+", "bench.source_hash " + H, l) for l in old[:header_lines]]
    (P / dst).write_text("\n".join(hdr) + ("\n\n" if header_lines > 1 else "\n") + body_path.read_text())


with_header("%s_instruction_budget.txt" % tag, G / ("%s_insts_summary.txt" % tag), 4)
with_header("%s_mix_timeline.txt" % tag, G / ("%s_mix_timeline.txt" % tag), 1)
with_header("%s_lone_proof.txt" % tag, G / ("%s_lone_proof.txt" % tag), 1)
print("profiles/%s_*: sources %s; value %.2f M, throughput %.2f ms per proof, lone proof %.1f ms of kernels in %d launches"
      % (tag, H, line["value"] / 1e6, line["throughput"]["ms_per_proof"], line["lone_proof"]["kernel_ms_sum"], line["lone_proof"]["launches"]))
