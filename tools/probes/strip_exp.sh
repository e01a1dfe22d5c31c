timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "d4x4 or fake_grid or strip" 2>&1 | tail -3
for v in "ITG_CONV_S2STRIP=1" "ITG_CONV_S2STRIP=0"; do echo "== $v"; env $v timeout -k 10 120 python tools/conv_bench.py "D0" 2>&1 | grep "D0"; done
