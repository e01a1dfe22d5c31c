#!/bin/bash
# The GPU test suite once per opt-in / fallback configuration: one line per variant into
# gpurun_out/<tag>_variants.txt (copied to profiles/pytest_gpu_<round>_variants.txt), full logs beside it.
# usage: tools/variant_suites.sh <tag> [first variant] [last variant]     (a suite takes ~2 min: at most 9 per 20-minute gpurun call)
TAG=${1:-variants}; FIRST=${2:-1}; LAST=${3:-99}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/${TAG}_variants.txt
[ $FIRST = 1 ] && : > $OUT
i=0
while IFS= read -r V; do
  i=$((i+1))
  if [ $i -lt $FIRST ] || [ $i -gt $LAST ]; then continue; fi
  env $V python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/${TAG}_variant_$i.log 2>&1
  echo "[$V] $(tail -1 gpurun_out/${TAG}_variant_$i.log)" | tee -a $OUT
  grep -E "^FAILED" gpurun_out/${TAG}_variant_$i.log | tee -a $OUT
done <<'VARS'
ITG_NONE=0
ITG_WINOGRAD=0
ITG_WINOGRAD_S2=0
ITG_DEFER_REDUCE=1
ITG_WINO_ACC64=2 ITG_STATS_PATHS=7
ITG_WINO_ACC64=3
ITG_WINO_ACC64=0
ITG_WINOGRAD_G=1 ITG_SN_FUSED_REDUCE=1
ITG_HALO_INTERIOR=1 ITG_OVERLAP=0
ITG_KERNEL_MASK=0xDFF
ITG_KERNEL_MASK=0
ITG_STATS_PATHS=5
VARS
