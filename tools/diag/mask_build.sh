#!/bin/bash
# Diagnostic build of the library with every gathered point index of the bucket sweep masked (BPG_DIAG_INDEX_MASK): the same instruction stream on a
# table that fits the L2 (0x3fff: 1.5 MB).  Proof bytes are WRONG by construction; only timings are read.  usage: mask_build.sh [mask=0x3fff]
# -> tools/diag/libbpg_hip_mask.so, used through BPG_LIB_PATH (bulletproofs_gadgets_amd/__init__.py) by tools/diag/mask_inflight.py
set -e
mask=${1:-0x3fff}
root=$(cd "$(dirname "$0")/../.." && pwd)
cd "$root/bulletproofs_gadgets_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed -Xarch_host -march=x86-64-v3 -DBPG_DIAG_INDEX_MASK=${mask}u \
    -o "$root/tools/diag/libbpg_hip_mask.so" engine.hip capi.hip
echo "built tools/diag/libbpg_hip_mask.so with index mask $mask"
