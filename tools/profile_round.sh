#!/bin/bash
# Evidence for profiles/: run on the GPU box from the repo root, e.g.  gpurun --timeout 1100 -- 'bash tools/profile_round.sh r01'
#   1. the default bench line (with cpu_baseline and in_flight)                      -> gpurun_out/<tag>_bench_plain.json
#   2. rocprofv3 --kernel-trace --stats of bench.py --headline-only and of plain bench.py -> gpurun_out/<tag>_stats/, <tag>_stats_default/  (+ bench lines under the profiler)
#   3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel trace only           -> gpurun_out/<tag>_pmc_fetch/, <tag>_pmc_write/
# rocprofv3 gets the program itself after "--" (python3 bench.py ...), never a shell or env wrapper.
set -o pipefail
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out
mkdir -p "$out"
timeout -k 10 400 python3 bench.py > "$out/${tag}_bench_plain.json" 2> "$out/${tag}_bench_plain.err" || { echo "bench failed"; tail -5 "$out/${tag}_bench_plain.err"; exit 1; }
echo "bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -o "$tag" -- python3 "$root/bench.py" --headline-only > "$out/${tag}_bench_under_rocprof.json" 2> "$out/${tag}_stats.err" || { echo "stats pass failed"; tail -5 "$out/${tag}_stats.err"; exit 1; }
echo "stats done"
# the same, for the default command line exactly as the driver runs it (verify / expanded-blinding / in-flight / CPU-baseline legs included:
# more launches of the same kernels, so only the fixed-size ones - the generator folds - keep the averages of the headline-only pass)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats_default" -o "$tag" -- python3 "$root/bench.py" > "$out/${tag}_bench_default_under_rocprof.json" 2> "$out/${tag}_stats_default.err" || { echo "default-command stats pass failed"; tail -5 "$out/${tag}_stats_default.err"; exit 1; }
echo "default-command stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_pmc_fetch" -o "$tag" -- python3 "$root/bench.py" --headline-only --steps 1 --warmup 1 > /dev/null 2> "$out/${tag}_pmc_fetch.err" || { echo "fetch pass failed"; tail -5 "$out/${tag}_pmc_fetch.err"; exit 1; }
echo "fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/${tag}_pmc_write" -o "$tag" -- python3 "$root/bench.py" --headline-only --steps 1 --warmup 1 > /dev/null 2> "$out/${tag}_pmc_write.err" || { echo "write pass failed"; tail -5 "$out/${tag}_pmc_write.err"; exit 1; }
echo "write done"
find "$out" -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head
