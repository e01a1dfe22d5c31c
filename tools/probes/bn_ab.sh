# BatchNorm kernels A/B on one box: round-3 norm.hip (ab_libs/libitg_oldnorm.so) vs this build, + the BN parity tests
set -e
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q -m gpu -k "bn or norm or ssm" 2>&1 | tail -3
echo "== old norm.hip"; ITG_LIB=$GRAFT_REPO_ROOT/ab_libs/libitg_oldnorm.so python tools/membound_bench.py 2>&1 | grep "BN train"
echo "== new norm.hip"; python tools/membound_bench.py 2>&1 | grep "BN train"
for B in 256 1024; do echo "== new, ITG_BN_RED_BLOCKS=$B"; ITG_BN_RED_BLOCKS=$B python tools/membound_bench.py 2>&1 | grep "BN train"; done
for B in 1024 4096; do echo "== new, ITG_BN_APPLY_BLOCKS=$B"; ITG_BN_APPLY_BLOCKS=$B python tools/membound_bench.py 2>&1 | grep "BN train"; done
echo "== new, 256-thread reductions only"; ITG_BN_RED_BIG=100000000 python tools/membound_bench.py 2>&1 | grep "BN train"
} > gpurun_out/r4f_bn_ab.log 2>&1
