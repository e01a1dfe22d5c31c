timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "strip or tile3x3 or halo_tile or statistics" 2>&1 | tail -3
for v in "ITG_STRIP_DEBUG=0"; do echo "== $v"; env $v timeout -k 10 120 python tools/conv_bench.py "b6c2" 2>&1 | grep "G b6c2"; env $v timeout -k 10 120 python tools/conv_bench.py "X tile b5c2" 2>&1 | grep "X tile"; done
ITG_STRIP_DEBUG=64 timeout -k 5 120 python tools/probes/strip_ts_probe.py 2>&1 | grep "strip ts" | tail -4
