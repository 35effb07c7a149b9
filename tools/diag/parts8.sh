#!/bin/bash
# first-fold scalars cut in eight (103 GB of tables) against four (51.5 GB), in the concurrent mix and for a proof alone
for cfg in "" "BPG_FOLD_PARTS=8 BPG_TABLE_GB=200 BPG_FOLD_TABLE_GB=128"; do
  a=$(env $cfg BPG_PROFILE=serving timeout -k 10 300 python3 tools/diag/kprof.py 512 3 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('alone %.2f ms (wnaf fold %.3f)' % (d['gpu_ms'], k['k_fold_points_wnaf'][1]))")
  b=$(env $cfg timeout -k 10 300 python3 bench.py --in-flight-only --in-flight-steps 48 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['in_flight']; print('sustained %.3f ms per proof' % d['ms_per_proof'])")
  echo "[$cfg] $a; $b"
done
