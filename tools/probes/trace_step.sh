# kernel trace of the overlapped config-1 step + launch-by-launch view of one step (tools/step_trace.py)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-trace}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -- python3 $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-direct > $OUT/${TAG}_trace.log 2>&1
python3 $ROOT/tools/step_trace.py $OUT/${TAG}_trace 3 --rows > $OUT/${TAG}_step_trace.txt
tail -12 $OUT/${TAG}_step_trace.txt
# keep the merged output small: drop the raw trace
rm -rf $OUT/${TAG}_trace
