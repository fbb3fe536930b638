#!/bin/bash
# per-kernel table of the fp32 HIP ViT at one small batch:  bash tools/vit_f32_small_prof.sh B OUT.csv
ROOT=${GRAFT_REPO_ROOT:-$PWD}
B=${1:-1}
OUT=${2:-$ROOT/gpurun_out/vit_f32_small_B$B.csv}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/vfs && SSLAM_SMALL_ONLY_DEFAULT=1 rocprofv3 --kernel-trace --stats -d /tmp/vfs -o x --output-format csv -- python $ROOT/tools/vit_f32_small_batch.py $B > /dev/null 2>&1
cp $(find /tmp/vfs -name '*kernel_stats.csv' | head -1) $OUT
python $ROOT/tools/kstats.py /tmp/vfs 16
