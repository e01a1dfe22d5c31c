#!/bin/bash
# The summaries of a round from one tools/profile_all.sh + profile_workload.sh call (tag <tag>, <tag>_c3/_c4/_c5):
#   tools/summarise_round.sh <tag> <round>     -> profiles/<round>_*   (run BEFORE touching csrc again: the traffic json
#   carries the kernel-source hash of the tree it is produced in)
set -e
TAG=${1:-r3prof}; R=${2:-r03}
cd "$(dirname "$0")/.."
python tools/prof_summary.py --stats gpurun_out/${TAG}_stats --iters 8 --fetch gpurun_out/${TAG}_fetch --write gpurun_out/${TAG}_write --round $R --cmd "python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-direct --no-membound (ITG_OVERLAP=0)" | tail -1
for wl in c3:config3 c4:config4 c5:config5; do t=${wl%%:*}; w=${wl##*:}
  python tools/prof_summary.py --stats gpurun_out/${TAG}_${t}_stats --iters 8 --fetch gpurun_out/${TAG}_${t}_fetch --write gpurun_out/${TAG}_${t}_write --round $R --suffix _$w --cmd "python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-membound (ITG_OVERLAP=0 ITG_GRAPH=0)" | tail -1
done
python tools/pmc_summary.py gpurun_out/${TAG}_pmc --round $R --steps 4 | tail -1
