# parity tests of the conv family + A/B of the bench against ab_libs/libitg_base.so on one box
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
TAG=${1:-ab}
{
python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu -k "conv or golden or step or stats" 2>&1 | tail -3
bash tools/ab_bench.sh "ITG_LIB=$GRAFT_REPO_ROOT/ab_libs/libitg_base.so" "ITG_X=1" 3 60 --no-direct
} > gpurun_out/${TAG}.log 2>&1
tail -12 gpurun_out/${TAG}.log
