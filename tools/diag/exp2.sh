cd $GRAFT_REPO_ROOT
for q in 4 8; do for f in 8 12 16 24; do
  GPU_MAX_HW_QUEUES=$q python bench.py --in-flight-only --in-flight $f --steps 4 2>/dev/null | sed "s/^/q$q f$f /" >> gpurun_out/exp2.txt
done; done
