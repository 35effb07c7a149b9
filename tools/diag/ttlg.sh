# where the table-driven tail starts (BPG_TT_LG: generators per side are frozen at 2^lg; 0 = never) against the headline, the kernels of one proof alone and the sustained rate
for lg in ${LGS:-14 13 12 11 10 9}; do
  BPG_TT_LG=$lg timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --batch 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['throughput'] if 'throughput' in d else d.get('in_flight')
print('TT_LG $lg: headline %.2f ms/step, kernels of one proof alone %.2f ms, sustained %.2f ms/proof' % (d['ms_per_step'], d['gpu_busy']['kernel_ms_per_proof'], t['ms_per_proof']))"
done
