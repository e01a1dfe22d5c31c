#!/bin/bash
# D's 256 -> 512 layer: direct weight gradient vs Winograd (ITG_WINOGRAD_WGRAD), then the train step either way
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
bash $ROOT/tools/ab_bench.sh "ITG_WINOGRAD_WGRAD=0" "ITG_WINOGRAD_WGRAD=1" ${1:-3} 60
