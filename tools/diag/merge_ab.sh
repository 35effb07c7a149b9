#!/bin/bash
# the equal-scalar merging (BPG_MERGE: 1 once per uploaded witness, 2 afresh in every proof, 0 never) on the reference's instance (512 equal leaves) and on a
# tree of distinct leaves: sustained ms per proof of the mix, alternating on one box
echo "== reference instance (512 equal leaves)"; bash tools/diag/knob_ab.sh "BPG_MERGE=2" "BPG_MERGE=0"
echo "== distinct leaves (--leaf-seed 7)"; BENCH_ARGS="--leaf-seed 7" bash tools/diag/knob_ab.sh "BPG_MERGE=2" "BPG_MERGE=0"
