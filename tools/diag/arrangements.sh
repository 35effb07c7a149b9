set -o pipefail
for cfg in "14 14 1 -" "20 1 1 14x1,1x6" "20 1 1 16x1,1x4" "20 1 1 10x2" "20 1 1 12x1,1x8" "18 18 1 -"; do
  set -- $cfg
  timeout -k 10 200 python3 bench.py --headline-only --steps 20 --warmup 5 --streams $1 --chain-workers $2 --chain-lanes $3 $( [ "$4" != "-" ] && echo --chain-pool $4 ) > gpurun_out/arr_$1_$2_$3_$4.json 2> gpurun_out/arr_$1_$2_$3_$4.err || { echo "FAILED $cfg"; tail -3 gpurun_out/arr_$1_$2_$3_$4.err; }
  python3 - <<PY
import json
d=json.load(open("gpurun_out/arr_$1_$2_$3_$4.json"))
print("streams $1 workers $2 lanes $3 pool $4: ms_per_step %.2f value %.1fM completions %s" % (d["ms_per_step"], d["value"]/1e6, {k: round(v) for k,v in d["completions"].items() if k!="note"}))
PY
done
nproc
