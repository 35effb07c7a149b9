# sustained rate and headline against the number of hardware queues (GPU_MAX_HW_QUEUES, read by the HIP runtime at start-up) and proving streams
for q in 4 8 12 16 24; do
  for n in 8 16; do
    GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 bench.py --in-flight-only --in-flight $n --in-flight-steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['in_flight']; print('queues $q streams $n: ms_per_proof %.2f value %.1fM' % (d['ms_per_proof'], d['value']/1e6))"
  done
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python3 bench.py --headline-only --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q headline: ms_per_step %.2f value %.1fM' % (d['ms_per_step'], d['value']/1e6))"
done
