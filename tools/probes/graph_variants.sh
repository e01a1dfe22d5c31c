#!/bin/bash
# schedule variants of the config-1 step on one box: eager / hipGraph replay, weight-gradient streams, nested fork, graph queues
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
run() { v=$(env $1 python3 $ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['value'])"); echo "$v  $1"; }
for r in 1 2; do
run "ITG_GRAPH=0"
run "ITG_GRAPH=0 ITG_WGRAD_STREAMS=1"
run "ITG_GRAPH=0 ITG_WGRAD_STREAMS=3"
run "ITG_GRAPH=0 ITG_NESTED_FORK=0"
run "ITG_GRAPH=0 ITG_DEFER_REDUCE=1"
run "ITG_GRAPH=0 ITG_SN_FUSED_REDUCE=1"
run "ITG_GRAPH=1"
run "ITG_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=4"
done
