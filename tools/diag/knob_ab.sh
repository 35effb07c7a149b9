#!/bin/bash
# sustained rate of the mix under a list of knob settings, alternating with the default, on one box
# usage: [BENCH_ARGS="--leaf-seed 7"] knob_ab.sh "VAR=val [VAR2=val]" ...      (each setting is run twice, interleaved with default runs)
run() { env $1 timeout -k 10 250 python3 bench.py --in-flight-only --in-flight-steps 48 $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['in_flight']; print('%.3f' % d['ms_per_proof'])"; }
for rep in 1 2; do
  echo "default: $(run BPG_NOOP=1) ms per proof"
  for kv in "$@"; do echo "$kv: $(run "$kv") ms per proof"; done
done
