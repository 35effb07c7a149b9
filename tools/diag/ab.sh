#!/bin/bash
# A/B of two builds of the library on the SAME box, alternating: sustained rate of the concurrent mix (bench.py --in-flight-only) per build.
# usage (GPU box, repo root): bash tools/diag/ab.sh <libA.so> <libB.so> [reps=2] [extra bench flags...]   -> gpurun_out/ab_<k>_{A,B}.json + a summary line per run
A=$1; B=$2; reps=${3:-2}; shift 3 2>/dev/null
mkdir -p gpurun_out
for k in $(seq 1 $reps); do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    BPG_LIB_PATH=$(realpath $lib) timeout -k 10 250 python3 bench.py --in-flight-only "$@" > gpurun_out/ab_${k}_$v.json 2> gpurun_out/ab_${k}_$v.err || { echo "run $k $v failed"; tail -3 gpurun_out/ab_${k}_$v.err; exit 1; }
    python3 -c "
import json,sys
d=json.loads(open('gpurun_out/ab_${k}_$v.json').read().strip().splitlines()[-1])['in_flight']
print('$k $v %-40s %.3f ms per proof sustained (%.3f whole run)' % ('$lib', d['ms_per_proof'], d['whole_run']['ms_per_proof']))"
  done
done
