# headline, kernels of one proof alone and sustained rate under a list of BPG_* settings; usage: sustained.sh "VAR=val VAR2=val" ...
for kv in "$@"; do
  env $kv timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --batch 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['throughput'] if 'throughput' in d else d.get('in_flight')
print('$kv: headline %.2f ms/step, kernels of one proof alone %.2f ms, sustained %.2f ms/proof' % (d['ms_per_step'], d['gpu_busy']['kernel_ms_per_proof'], t['ms_per_proof']))"
done
