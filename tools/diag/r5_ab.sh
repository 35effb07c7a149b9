#!/bin/bash
# round 5: sustained rate of the mix under the new knobs against their round-4 settings, alternating on one box (tools/diag/knob_ab.sh does the runs)
bash tools/diag/knob_ab.sh "$@"
