#!/bin/bash
# Stall-reason counters for the MFMA kernels (run on the GPU box from the repo root): two --pmc passes, summarised to gpurun_out/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
RX='gather_refine|selector_saliency|sim_argmax|refine_bf16|selector_bf16'
ARGS="$ROOT/bench.py --steps 2 --warmup 1 --no-vit --no-cpu-baseline"
rm -rf /tmp/pst1 /tmp/pst2
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM \
  --kernel-trace --kernel-include-regex "$RX" -d /tmp/pst1 -o x --output-format csv -- python $ARGS > /dev/null
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES \
  --kernel-trace --kernel-include-regex "$RX" -d /tmp/pst2 -o x --output-format csv -- python $ARGS > /dev/null
python $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmc_stall.json /tmp/pst1 /tmp/pst2 > /dev/null
