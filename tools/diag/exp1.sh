set -x
cd $GRAFT_REPO_ROOT
python tools/diag/kprof.py 512 3 0 > gpurun_out/k_base.json 2>gpurun_out/k_base.err || exit 1
python tools/diag/kprof.py 512 3 1 > gpurun_out/k_prefetch.json 2>>gpurun_out/k_base.err || exit 1
BPG_TILE_LGMAX=17 BPG_TILE_SHIFT=3 BPG_TILE_THREADS=1024 python tools/diag/kprof.py 512 3 0 > gpurun_out/k_tile17.json 2>>gpurun_out/k_base.err || exit 1
BPG_TILE_LGMAX=16 BPG_TILE_SHIFT=4 BPG_TILE_THREADS=1024 python tools/diag/kprof.py 512 3 0 > gpurun_out/k_tile16.json 2>>gpurun_out/k_base.err || exit 1
BPG_TILE_LGMAX=16 BPG_TILE_SHIFT=4 BPG_TILE_THREADS=512 python tools/diag/kprof.py 512 3 0 > gpurun_out/k_tile16_512.json 2>>gpurun_out/k_base.err || exit 1
BPG_RSEG=16 python tools/diag/kprof.py 512 3 0 > gpurun_out/k_rseg16.json 2>>gpurun_out/k_base.err || exit 1
BPG_MSM_CMAX=14 python tools/diag/kprof.py 512 3 0 > gpurun_out/k_cmax14.json 2>>gpurun_out/k_base.err || exit 1
python bench.py --in-flight-only --in-flight 12 --steps 4 > gpurun_out/if_q4.json 2>>gpurun_out/k_base.err || exit 1
GPU_MAX_HW_QUEUES=8 python bench.py --in-flight-only --in-flight 12 --steps 4 > gpurun_out/if_q8.json 2>>gpurun_out/k_base.err || exit 1
GPU_MAX_HW_QUEUES=16 python bench.py --in-flight-only --in-flight 12 --steps 4 > gpurun_out/if_q16.json 2>>gpurun_out/k_base.err || exit 1
