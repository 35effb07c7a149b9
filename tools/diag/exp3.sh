cd $GRAFT_REPO_ROOT
rm -f gpurun_out/exp3.txt
for w in 1 2 4 8 10 12; do
  python bench.py --headline-only --steps 24 --warmup 3 --chain-workers $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('workers $w', 'ms_per_step', round(d['ms_per_step'],2), 'value', round(d['value']/1e6,2), 'sweep avg', round(d['roofline']['avg_launch_ms'],4))" >> gpurun_out/exp3.txt
done
cat gpurun_out/exp3.txt
