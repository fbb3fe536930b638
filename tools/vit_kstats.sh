#!/bin/bash
# per-kernel times of the HIP ViT alone (rocprofv3 --kernel-trace --stats): bash tools/vit_kstats.sh <tag> <chunk>
# (SSLAM_BENCH_VIT=fp32 in the environment: the fp32-operand ViT)
TAG=${1:-vit}; CHUNK=${2:-41}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kv && rocprofv3 --kernel-trace --stats -d /tmp/kv -o x --output-format csv -- python $ROOT/tools/bench_vit.py 448 $CHUNK > /dev/null 2>&1
cp $(find /tmp/kv -name '*kernel_stats.csv' | head -1) $ROOT/gpurun_out/${TAG}_vit_kernel_stats.csv
python $ROOT/tools/kstats.py /tmp/kv 14
