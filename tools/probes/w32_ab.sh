# blocked accumulation of the Winograd forward GEMMs: block sums in fp64 (NT_W64) vs in a second fp32 accumulator (NT_W32)
set -o pipefail
for m in 1 4; do
ITG_WINO_ACC64=$m python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "winograd" -s 2>&1 | grep -E "rel-L2|passed|failed"
ITG_WINO_ACC64=$m python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config2 and winograd" -s 2>&1 | grep -E "G gradients|passed|failed|assert"
done
bash tools/ab_bench.sh "ITG_WINO_ACC64=1" "ITG_WINO_ACC64=4" 3 60 --no-direct --no-membound 2>&1 | tail -1
