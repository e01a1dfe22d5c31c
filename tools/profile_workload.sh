#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes + kernel stats of one bench workload (config3 | config4), un-overlapped, eager.
# usage: tools/profile_workload.sh <tag> <workload>     -> gpurun_out/<tag>_{stats,fetch,write}
set -e
TAG=$1; WL=$2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export ITG_OVERLAP=0 ITG_GRAPH=0
CMD="python3 $ROOT/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-membound"      # (--no-membound: rounds 2-5 profiled the memory-bound operator probes along with the step - their local_pad / upsample / BatchNorm launches were read as the step's)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- $CMD > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -- $CMD > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/${TAG}_write -- $CMD > $OUT/${TAG}_write.log 2>&1
echo done $WL
