#!/bin/bash
# Round 6: un-split / small-tile plans for the generator's wide layers.  The two planner overrides it sets were
# experiment knobs (removed once the numbers were in DESIGN section 3): kept as the record of what was swept.
# experimental planner overrides ITG_X_NT="bco,bpix,ks" (forward / input gradient) and ITG_X_TN="bcol,bco,splits" (weight gradient).
out=${1:-gpurun_out/r6_plans.log}
: > $out
for v in "" "32,64,1" "64,64,1" "32,128,1" "64,128,1" "112,64,1" "112,128,1" "64,64,2" "64,64,3" "32,64,2" "64,128,2" "112,64,3" "112,64,5"; do
  echo "== ITG_X_NT=$v" >> $out
  ITG_X_NT="$v" python tools/conv_bench.py XW 2>&1 | grep -v "^sum" | sed 's/| wgrad.*//' >> $out
done
for v in "" "64,64,1" "64,64,2" "64,64,3" "128,128,1" "128,128,2" "128,128,3" "128,128,4" "256,64,1" "256,64,2" "256,64,3"; do
  echo "== ITG_X_TN=$v" >> $out
  ITG_X_TN="$v" python tools/conv_bench.py XW 2>&1 | grep -v "^sum" | sed 's/| fwd.*| wgrad/| wgrad/' >> $out
done
