#!/bin/bash
# config 4 on one rank: the bench line + a kernel-trace statistics pass (un-overlapped)  ->  gpurun_out/<tag>_c4*
TAG=${1:-c4}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT && python bench.py --workload config4 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_c4_bench.json 2> $OUT/${TAG}_c4_bench.err
cd /tmp && export TMPDIR=/tmp
ITG_OVERLAP=0 ITG_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c4_stats -- python3 $ROOT/bench.py --workload config4 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_c4_stats.log 2>&1
