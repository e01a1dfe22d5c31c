# per-kernel stats of one un-overlapped config-1 bench (the first pass of tools/profile_all.sh)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-st}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export ITG_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-direct --no-membound > $OUT/${TAG}_stats.log 2>&1
python3 $ROOT/tools/kstats.py $OUT/${TAG}_stats > $OUT/${TAG}_kstats.txt
find $OUT/${TAG}_stats -name "*kernel_trace.csv" -delete
head -70 $OUT/${TAG}_kstats.txt
