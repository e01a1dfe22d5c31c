#!/bin/bash
# Every rocprofv3 pass behind the committed summaries of a round, in one GPU call:
#   kernel-trace + stats of the bench (kernels one at a time: ITG_OVERLAP=0), FETCH_SIZE and WRITE_SIZE passes of the same
#   command (separate --pmc passes as MI355X_MICROARCH.md prescribes), the two SQ counter passes (tools/pmc_run.sh).
# usage: tools/profile_all.sh <tag>       -> gpurun_out/<tag>_{stats,fetch,write,pmc_A,pmc_B}
set -e
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export ITG_OVERLAP=0
CMD="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-direct --no-membound"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- $CMD > $OUT/${TAG}_stats.log 2>&1
echo stats done
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -- $CMD > $OUT/${TAG}_fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/${TAG}_write -- $CMD > $OUT/${TAG}_write.log 2>&1
echo write done
unset ITG_OVERLAP
bash $ROOT/tools/pmc_run.sh ${TAG}_pmc config1 > /dev/null
echo pmc done
