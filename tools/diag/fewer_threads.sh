# fewer proving threads: each stream keeps TWO chains in flight on the shared pool (its current proof's and its next one's), so ten streams start twenty chains at once
for arg in "--streams 20" "--streams 10 --chain-pool 13x1,1x7" "--streams 8 --chain-pool 13x1,1x7" "--streams 12 --chain-pool 13x1,1x7" "--streams 10 --chain-pool 12x1,1x8" "--streams 7 --chain-pool 13x1,1x7"; do
  timeout -k 10 200 python3 bench.py --headline-only --steps 20 --warmup 5 $arg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['config']['host_threads_per_gpu']
print('$arg: %.2f ms/step  %.1f M  chain threads %s proving %s' % (d['ms_per_step'], d['value']/1e6, h.get('chain_workers'), h.get('proving')))"
done
