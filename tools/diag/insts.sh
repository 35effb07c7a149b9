#!/bin/bash
# VALU instruction budget of a proof, per kernel: rocprofv3 --pmc pass (kernel trace only) over tools/diag/insts_workload.py
# usage (GPU box, repo root): bash tools/diag/insts.sh <tag> [VAR=val ...]      -> gpurun_out/<tag>_insts/ + summary on stdout
set -o pipefail
tag=$1; shift
root=$(pwd); out=$root/gpurun_out; mkdir -p "$out"; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$out/${tag}_insts" -o "$tag" -- python3 "$root/tools/diag/insts_workload.py" 512 2 > "$out/${tag}_insts.log" 2> "$out/${tag}_insts.err") || { echo "insts pass failed"; tail -5 "$out/${tag}_insts.err"; exit 1; }
cc=$(find "$out/${tag}_insts" -name "*counter_collection.csv" | head -1); kt=$(find "$out/${tag}_insts" -name "*kernel_trace.csv" | head -1)
python3 "$root/tools/diag/insts_summarize.py" "$cc" 2 "$kt" | tee "$out/${tag}_insts_summary.txt"
cp "$cc.summary.json" "$out/${tag}_insts_summary.json"
rm -rf "$out/${tag}_insts"          # the raw CSVs are tens of MB
