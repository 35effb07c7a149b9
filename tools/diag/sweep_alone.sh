#!/bin/bash
# the bucket sweep of one proof ALONE on the device under a list of knob settings (HIP events of every kernel: bench.py --kernel-profile, one stream)
# usage: sweep_alone.sh "VAR=val ..." ...
for kv in "BPG_NOOP=1" "$@"; do
  env $kv timeout -k 10 200 python3 bench.py --headline-only --streams 1 --chain-workers 1 --steps 3 --warmup 1 --kernel-profile $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms',{})
tot=sum(k.values())
print('%-40s all kernels %.2f ms | ' % ('$kv', tot) + ', '.join('%s %.2f' % (n, v) for n, v in k.items() if 'bucket' in n or 'window' in n))"
done
