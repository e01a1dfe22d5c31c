#!/bin/bash
# The four bench workloads back to back on one box -> gpurun_out/<tag>_{config1,config3,config4,config5}.json
# usage: tools/bench_all.sh <tag>
TAG=${1:-ball}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
python3 bench.py > gpurun_out/${TAG}_config1.json 2> gpurun_out/${TAG}_config1.err && echo c1 done &&
python3 bench.py --workload config3 --no-cpu-baseline > gpurun_out/${TAG}_config3.json 2> gpurun_out/${TAG}_config3.err && echo c3 done &&
python3 bench.py --workload config4 --no-cpu-baseline > gpurun_out/${TAG}_config4.json 2> gpurun_out/${TAG}_config4.err && echo c4 done &&
python3 bench.py --workload config5 --no-cpu-baseline > gpurun_out/${TAG}_config5.json 2> gpurun_out/${TAG}_config5.err && echo c5 done
