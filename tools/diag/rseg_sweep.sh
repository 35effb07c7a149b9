#!/bin/bash
# BPG_RSEG (buckets per thread of the first epilogue level) against kernel time of a proof alone and the sustained rate of the mix
for r in 4 8 16 32; do
  a=$(BPG_RSEG=$r BPG_PROFILE=serving timeout -k 10 200 python3 tools/diag/kprof.py 512 3 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('alone %.2f ms (reduce %.3f window_sums %.3f combine %.3f)' % (d['gpu_ms'], k['k_bucket_reduce'][1], k['k_window_sums'][1], k.get('k_bucket_combine',[0,0])[1]))")
  b=$(BPG_RSEG=$r timeout -k 10 250 python3 bench.py --in-flight-only --in-flight-steps 48 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['in_flight']; print('sustained %.3f ms per proof' % d['ms_per_proof'])")
  echo "BPG_RSEG=$r: $a; $b"
done
