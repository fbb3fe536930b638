#!/bin/bash
# Per-kernel average durations of the bf16 ViT forward at 82 frames, for a variant build of the library (or the in-tree one):
#   tools/vit_bf16_kernel_times.sh TAG [variant.so]      -> gpurun_out/ktb_TAG.txt
TAG=$1; [ -n "$2" ] && export SSLAM_BENCH_SO=$2
cd "$(dirname "$0")/.." && export TMPDIR=/tmp && mkdir -p gpurun_out
rm -rf /tmp/ktb_$TAG && rocprofv3 --kernel-trace --stats -d /tmp/ktb_$TAG -o p --output-format csv -- python3 tools/bench_vit.py 448 82 > gpurun_out/ktb_$TAG.txt 2>&1
python3 - "$TAG" >> gpurun_out/ktb_$TAG.txt <<'PY'
import csv, glob, sys
f = glob.glob(f"/tmp/ktb_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:5]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us")
PY
grep "frames/s\| us$" gpurun_out/ktb_$TAG.txt
