#!/bin/bash
# The headline's appetite as a measured trade-off: value, HBM in use and busy host cores of the SAME 20 timed steps for
# {serving, one-shot} x {20, 10, 6} proving streams.  usage (GPU box, repo root): bash tools/diag/appetite.sh > gpurun_out/appetite.txt
echo "profile   streams  chain_threads  ms/step   M constraints/s  HBM in use GB  busy cores  first/median/last completion ms"
for prof in serving oneshot; do
  for streams in 20 10 6; do
    out=gpurun_out/appetite_${prof}_${streams}.json
    timeout -k 10 300 python3 bench.py --headline-only --no-one-shot-leg --profile $prof --streams $streams --steps 20 --warmup 5 > $out 2> ${out%.json}.err || { echo "$prof $streams failed"; tail -3 ${out%.json}.err; continue; }
    python3 - "$out" "$prof" "$streams" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = d["config"]["host_threads_per_gpu"]; c = d.get("completions") or {}
print("%-9s %7s  %13s  %7.2f  %15.1f  %13.1f  %10.1f  %s" % (sys.argv[2], sys.argv[3], h["chain_workers"], d["ms_per_step"], d["value"] / 1e6,
      (d.get("hbm_in_use") or {}).get("in_use_GB", float("nan")), h["busy_cores_avg"], "%.0f / %.0f / %.0f" % (c.get("first_ms", 0), c.get("median_ms", 0), c.get("last_ms", 0))))
PY
  done
done
