set -e
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
{
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_model.py -x -q -m gpu -k "bn or norm or ssm or golden" 2>&1 | tail -3
python tools/membound_bench.py 2>&1 | grep -v amdgpu.ids
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4i_bench.json 2> gpurun_out/r4i_bench.err
python -c "
import json;d=json.load(open('gpurun_out/r4i_bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['direct_algorithm']);[print(r) for r in d['roofline']['membound']]"
} > gpurun_out/r4i_bn_check.log 2>&1
