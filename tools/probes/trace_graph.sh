# kernel trace of the hipGraph-replayed config-1 step (no host effects): timeline + launch-by-launch view
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-tg}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export ITG_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -- python3 $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-direct > $OUT/${TAG}_trace.log 2>&1
python3 $ROOT/tools/timeline.py $OUT/${TAG}_trace > $OUT/${TAG}_timeline.txt 2>&1
python3 $ROOT/tools/step_trace.py $OUT/${TAG}_trace 3 --rows > $OUT/${TAG}_step_trace.txt 2>&1
cat $OUT/${TAG}_timeline.txt; tail -8 $OUT/${TAG}_step_trace.txt
rm -rf $OUT/${TAG}_trace
