# A/B of the Winograd GEMMs' accumulation (round 4): plain fp32 chain vs blocked fp64 accumulation, block 16 / 8 / 4
set -e
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== ACC64 off"; ITG_WINO_ACC64=0 python tools/wino_accuracy.py
echo "== ACC64 b16 (default)"; python tools/wino_accuracy.py
echo "== ACC64 b8"; ITG_LIB=$GRAFT_REPO_ROOT/ab_libs/libitg_b8.so python tools/wino_accuracy.py
echo "== ACC64 b4"; ITG_LIB=$GRAFT_REPO_ROOT/ab_libs/libitg_b4.so python tools/wino_accuracy.py
} > gpurun_out/r4a_wino_acc.log 2>&1
