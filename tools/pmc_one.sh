#!/bin/bash
# Counters of one kernel family under bench.py (two --pmc passes; kernel trace only):  bash tools/pmc_one.sh <regex> <tag>
ROOT=${GRAFT_REPO_ROOT:-$PWD}
RX=${1:-sim_argmax}; TAG=${2:-pmc_one}
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 2 --warmup 1 --no-vit --no-bf16 --no-cpu-baseline"
rm -rf /tmp/po1 /tmp/po2 /tmp/po3
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU \
  --kernel-trace --kernel-include-regex "$RX" -d /tmp/po1 -o x --output-format csv -- python $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM \
  --kernel-trace --kernel-include-regex "$RX" -d /tmp/po2 -o x --output-format csv -- python $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE \
  --kernel-trace --kernel-include-regex "$RX" -d /tmp/po3 -o x --output-format csv -- python $ARGS > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/${TAG}.json /tmp/po1 /tmp/po2 /tmp/po3
