cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
{
python -m pytest tests/test_gpu_cli.py tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -5
echo "== cli graph (auto)"; python tools/cli_throughput.py 2>&1 | grep -v amdgpu.ids | tail -4
echo "== cli eager"; python tools/cli_throughput.py 2400 --launch_mode eager 2>&1 | grep -v amdgpu.ids | tail -4
} > gpurun_out/r4m.log 2>&1
