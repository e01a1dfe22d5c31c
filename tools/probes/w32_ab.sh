# blocked accumulation of the Winograd forward GEMMs: block sums in fp64 (NT_W64), in a second fp32 accumulator (NT_W32),
# in a compensated fp32 sum (NT_W32K).  ITG_WINO_ACC64 = 1 default (F(4,4) fp64, F(4,2) fp32), 3 / 4 / 5 = all fp64 / fp32 / Kahan
set -o pipefail
for m in 3 4 5; do
echo "== ITG_WINO_ACC64=$m"
ITG_WINO_ACC64=$m python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "winograd" -s 2>&1 | grep -E "rel-L2|passed|failed"
done
for m in 1 3 4 5 1 3 4 5; do
ITG_WINO_ACC64=$m python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-direct --no-membound | python3 -c "import json,sys; print('mode $m', json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"
done
