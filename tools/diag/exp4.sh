cd $GRAFT_REPO_ROOT
rm -f gpurun_out/exp4.txt
for cfg in "1 10" "2 10" "2 12" "3 12" "4 12"; do
  set -- $cfg
  python bench.py --headline-only --steps 20 --warmup 5 --streams $1 --chain-workers $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $1 workers $2', 'ms_per_step', round(d['ms_per_step'],2), 'value', round(d['value']/1e6,2), d['config']['host_threads_per_gpu'])" >> gpurun_out/exp4.txt
done
cat gpurun_out/exp4.txt
