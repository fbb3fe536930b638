#!/bin/bash
# per-kernel table of the bf16 HIP ViT at one small batch:  bash tools/vit_bf16_small_prof.sh B
ROOT=${GRAFT_REPO_ROOT:-$PWD}
B=${1:-1}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/vbs && rocprofv3 --kernel-trace --stats -d /tmp/vbs -o x --output-format csv -- python $ROOT/tools/bench_vit.py 448 $B $B $B $B $B $B > /dev/null 2>&1
python $ROOT/tools/kstats.py /tmp/vbs 12
