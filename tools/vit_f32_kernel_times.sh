#!/bin/bash
# Per-kernel average durations of the fp32 ViT forward at FRAMES (default 83) frames, for a variant build of the library (or the in-tree one):
#   tools/vit_f32_kernel_times.sh TAG [variant.so]      -> gpurun_out/kt_TAG.txt
TAG=$1; export SSLAM_BENCH_VIT=fp32; [ -n "$2" ] && export SSLAM_BENCH_SO=$2
cd "$(dirname "$0")/.." && export TMPDIR=/tmp && mkdir -p gpurun_out
rm -rf gpurun_out/kt_$TAG && rocprofv3 --kernel-trace --stats -d gpurun_out/kt_$TAG -o p --output-format csv -- python3 tools/bench_vit.py 448 ${FRAMES:-83} > gpurun_out/kt_$TAG.txt 2>&1
python3 - "$TAG" >> gpurun_out/kt_$TAG.txt <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/kt_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us")
PY
rm -rf gpurun_out/kt_$TAG; grep -v "amdgpu.ids\|simple_timer\|generateRocpd\|tool.cpp" gpurun_out/kt_$TAG.txt | tail -9
