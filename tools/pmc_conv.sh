#!/bin/bash
# PMC passes over tools/conv_bench.py for one layer filter: where a single conv kernel spends its cycles.
#   tools/pmc_conv.sh <tag> "<filter>"   -> gpurun_out/<tag>_{A,B}; summarise with tools/pmc_summary.py gpurun_out/<tag> --round rXX
TAG=${1:-pmcc}
FLT=${2:-G b6c1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_A \
  --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA \
  -- python3 $ROOT/tools/conv_bench.py "$FLT" > $ROOT/gpurun_out/${TAG}_A.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_B \
  --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES \
  -- python3 $ROOT/tools/conv_bench.py "$FLT" > $ROOT/gpurun_out/${TAG}_B.log 2>&1
