#!/bin/bash
# one 2^20 proof alone on the device under {one-shot, serving} x {blocking, spinning} stream waits (tools/diag/lone_proof.py, one process each)
for p in oneshot serving; do for w in blocking spin; do timeout -k 10 200 python3 tools/diag/lone_proof.py $p $w 5 "$@" 2>/dev/null || echo "{\"$p/$w\": \"failed\"}"; done; done
