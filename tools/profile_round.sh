#!/bin/bash
# Evidence for profiles/: run on the GPU box from the repo root, e.g.
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r02 calib stats pmc'      and      gpurun --timeout 900 -- 'bash tools/profile_round.sh r02 bench'
# stages (any subset, in this order):
#   calib  FETCH_SIZE calibration on a known byte count (tools/calib/gather_calib.hip)          -> gpurun_out/<tag>_fetch_calibration.json
#   stats  rocprofv3 --kernel-trace --stats of bench.py --headline-only (as timed: 14 proving streams), of the same with one stream and
#          one chain thread (kernels alone on the GPU), and of --in-flight-only  -> gpurun_out/<tag>_stats/, <tag>_stats_single/, <tag>_stats_inflight/
#   pmc    two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel trace only                -> gpurun_out/<tag>_pmc_fetch/, <tag>_pmc_write/, <tag>_pmc_traffic.json
#   pmcthr the same two --pmc passes over `bench.py --in-flight-only` (the concurrent mix of the throughput leg)               -> gpurun_out/<tag>_pmc_traffic_throughput.json
#   bench  the default bench line exactly as the driver runs it (cpu_baseline, throughput, batch) -> gpurun_out/<tag>_bench_plain.json
#   insts  VALU instruction budget of a proof per kernel under the shared-device variants (tools/diag/insts.sh: one --pmc pass, SQ counters) -> <tag>_insts_summary.{txt,json}
#   mix    rocprofv3 kernel trace of `bench.py --in-flight-only` -> gap analysis (tools/diag/mix_timeline.py) -> <tag>_mix_timeline.txt
#   mask   the concurrent mix with every gathered point index masked to an L2-resident table against the product build, alternating
#          (tools/diag/mask_build.sh must have built tools/diag/libbpg_hip_mask.so in the build container) -> <tag>_mask_ab.txt
#   clock  shader clock and board power while the mix / one stream runs (tools/diag/clock_watch.py) -> <tag>_clock.txt
#   appetite  value / HBM / busy cores for {serving, one-shot} x {20, 10, 6} streams (tools/diag/appetite.sh) -> <tag>_appetite.txt
#   lone   one 2^20 proof ALONE on the device under both profiles and both kinds of stream wait (tools/diag/lone_proof.sh) -> <tag>_lone_proof.txt
#   inv    what a field inversion and a mixed addition cost in field multiplications (tools/diag/bench_inv.hip): the constants behind "no batched-affine sweep" -> <tag>_inversion_cost.json
# Back in the build container: python tools/collect_round.py <tag> copies the files into profiles/ (and refuses evidence of other sources); then
# the sanitizer evidence (profiles/<tag>_sanitizers.txt) is taken in the BUILD container, last, by tools/sanitize_round.sh <tag>, which fails when its source
# hash differs from the one in the counter files this script wrote.
# rocprofv3 gets the program itself after "--" (python3 bench.py ... or the calibration binary), never a shell or env wrapper.
set -o pipefail
tag=${1:-r05}; shift
stages=${*:-calib stats pmc bench}
root=$(pwd)
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
# read by the HIP runtime when it initialises - under rocprofv3 that is BEFORE python starts (the profiler's preloaded library touches the GPU first), so
# bench.py's os.environ.setdefault would come too late and the profiled runs would use 4 hardware queues where the timed line uses 8
export GPU_MAX_HW_QUEUES=8
has() { [[ " $stages " == *" $1 "* ]]; }
csv() { find "$1" -name "*$2" | head -1; }

if has calib; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_calib "$root/tools/calib/gather_calib.hip" || { echo "calib build failed"; exit 1; }
    (cd /tmp && timeout -k 10 120 /tmp/gather_calib > "$out/${tag}_calib_known.json") || { echo "calib run failed"; exit 1; }
    (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_calib_pmc" -o "$tag" -- /tmp/gather_calib > /dev/null 2> "$out/${tag}_calib_pmc.err") || { echo "calib pmc pass failed"; tail -5 "$out/${tag}_calib_pmc.err"; exit 1; }
    python3 "$root/tools/calib/fetch_factor.py" "$out/${tag}_calib_known.json" "$(csv "$out/${tag}_calib_pmc" counter_collection.csv)" "$out/${tag}_fetch_calibration.json" || exit 1
    echo "calib done"
fi
if has inv; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -o /tmp/bench_inv "$root/tools/diag/bench_inv.hip" || { echo "bench_inv build failed"; exit 1; }
    (cd /tmp && timeout -k 10 120 /tmp/bench_inv > "$out/${tag}_inversion_cost.json") || { echo "bench_inv run failed"; exit 1; }
    echo "inv done"
fi
if has stats; then
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -o "$tag" -- python3 "$root/bench.py" --headline-only --steps 20 --warmup 5 > "$out/${tag}_bench_under_rocprof.json" 2> "$out/${tag}_stats.err") || { echo "stats pass failed"; tail -5 "$out/${tag}_stats.err"; exit 1; }
    echo "stats done"
    # the same kernels alone on the GPU: one proving stream, one chain thread (the `isolated` figures of the bench line, the round-1 command)
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats_single" -o "$tag" -- python3 "$root/bench.py" --headline-only --streams 1 --chain-workers 1 --steps 5 --warmup 2 > "$out/${tag}_bench_single_under_rocprof.json" 2> "$out/${tag}_stats_single.err") || { echo "single-stream stats pass failed"; tail -5 "$out/${tag}_stats_single.err"; exit 1; }
    echo "single-stream stats done"
    # the concurrent mix of the throughput leg (6 proving streams, one chain thread with 8 lanes each, 24 proofs per stream): kernels of different proofs share the CUs
    (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats_inflight" -o "$tag" -- python3 "$root/bench.py" --in-flight-only > "$out/${tag}_bench_inflight_under_rocprof.json" 2> "$out/${tag}_stats_inflight.err") || { echo "in-flight stats pass failed"; tail -5 "$out/${tag}_stats_inflight.err"; exit 1; }
    echo "in-flight stats done"
fi
if has pmc; then
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_pmc_fetch" -o "$tag" -- python3 "$root/bench.py" --headline-only --streams 1 --chain-workers 1 --steps 1 --warmup 1 > /dev/null 2> "$out/${tag}_pmc_fetch.err") || { echo "fetch pass failed"; tail -5 "$out/${tag}_pmc_fetch.err"; exit 1; }
    echo "fetch done"
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/${tag}_pmc_write" -o "$tag" -- python3 "$root/bench.py" --headline-only --streams 1 --chain-workers 1 --steps 1 --warmup 1 > /dev/null 2> "$out/${tag}_pmc_write.err") || { echo "write pass failed"; tail -5 "$out/${tag}_pmc_write.err"; exit 1; }
    echo "write done"
    calib="$out/${tag}_fetch_calibration.json"; [ -f "$calib" ] || calib="$root/profiles/${tag}_fetch_calibration.json"; [ -f "$calib" ] || calib="$root/profiles/r02_fetch_calibration.json"
    python3 "$root/tools/pmc_summarize.py" "$(csv "$out/${tag}_pmc_fetch" counter_collection.csv)" "$(csv "$out/${tag}_pmc_write" counter_collection.csv)" "$calib" "$out/${tag}_pmc_traffic.json" || exit 1
fi
if has pmcthr; then
    # the same two counters over the CONCURRENT kernel mix of the throughput leg (6 proving streams): what a proof costs in HBM bytes when the GPU is shared
    (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_pmcthr_fetch" -o "$tag" -- python3 "$root/bench.py" --in-flight-only --in-flight-steps 8 > "$out/${tag}_pmcthr_fetch.json" 2> "$out/${tag}_pmcthr_fetch.err") || { echo "throughput fetch pass failed"; tail -5 "$out/${tag}_pmcthr_fetch.err"; exit 1; }
    echo "throughput fetch done"
    (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/${tag}_pmcthr_write" -o "$tag" -- python3 "$root/bench.py" --in-flight-only --in-flight-steps 8 > "$out/${tag}_pmcthr_write.json" 2> "$out/${tag}_pmcthr_write.err") || { echo "throughput write pass failed"; tail -5 "$out/${tag}_pmcthr_write.err"; exit 1; }
    echo "throughput write done"
    calib="$out/${tag}_fetch_calibration.json"; [ -f "$calib" ] || calib="$root/profiles/${tag}_fetch_calibration.json"; [ -f "$calib" ] || calib="$root/profiles/r02_fetch_calibration.json"
    python3 "$root/tools/pmc_summarize.py" "$(csv "$out/${tag}_pmcthr_fetch" counter_collection.csv)" "$(csv "$out/${tag}_pmcthr_write" counter_collection.csv)" "$calib" "$out/${tag}_pmc_traffic_throughput.json" "$out/${tag}_pmcthr_fetch.json" || exit 1
fi
if has bench; then
    timeout -k 10 800 python3 bench.py --steps 20 --warmup 5 > "$out/${tag}_bench_plain.json" 2> "$out/${tag}_bench_plain.err" || { echo "bench failed"; tail -5 "$out/${tag}_bench_plain.err"; exit 1; }
    echo "bench done"
fi
if has insts; then
    bash "$root/tools/diag/insts.sh" "$tag" BPG_PROFILE=serving BPG_FOLD_ADAPT=2 > "$out/${tag}_insts.log" 2>&1 || { echo "insts failed"; tail -5 "$out/${tag}_insts.log"; exit 1; }
    echo "insts done"
fi
if has mix; then
    (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$out/${tag}_mixtrace" -o "$tag" -- python3 "$root/bench.py" --in-flight-only --in-flight-steps 32 > "$out/${tag}_mixtrace.json" 2> "$out/${tag}_mixtrace.err") || { echo "mix trace failed"; tail -5 "$out/${tag}_mixtrace.err"; exit 1; }
    python3 "$root/tools/diag/mix_timeline.py" "$(csv "$out/${tag}_mixtrace" kernel_trace.csv)" 2.0 > "$out/${tag}_mix_timeline.txt" || exit 1
    rm -rf "$out/${tag}_mixtrace"
    echo "mix done"
fi
if has mask; then
    [ -f "$root/tools/diag/libbpg_hip_mask.so" ] || { echo "tools/diag/libbpg_hip_mask.so missing: run tools/diag/mask_build.sh in the build container"; exit 1; }
    bash "$root/tools/diag/ab.sh" "$root/bulletproofs_gadgets_amd/libbpg_hip.so" "$root/tools/diag/libbpg_hip_mask.so" 3 --in-flight-steps 48 > "$out/${tag}_mask_ab.txt" 2>&1 || { echo "mask A/B failed"; tail -5 "$out/${tag}_mask_ab.txt"; exit 1; }
    echo "mask done"
fi
if has clock; then
    python3 "$root/tools/diag/clock_watch.py" "mix (6 proving streams)" -- python3 "$root/bench.py" --in-flight-only --in-flight-steps 48 > "$out/${tag}_clock.txt" 2> "$out/${tag}_clock.err" || true
    python3 "$root/tools/diag/clock_watch.py" "one stream" -- python3 "$root/bench.py" --headline-only --streams 1 --chain-workers 1 --steps 5 --warmup 2 >> "$out/${tag}_clock.txt" 2>> "$out/${tag}_clock.err" || true
    echo "clock done"
fi
if has lone; then
    bash "$root/tools/diag/lone_proof.sh" > "$out/${tag}_lone_proof.txt" 2>&1 || true
    echo "lone done"
fi
if has appetite; then
    bash "$root/tools/diag/appetite.sh" > "$out/${tag}_appetite.txt" 2>&1 || true
    echo "appetite done"
fi
find "$out" -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head
