#!/bin/bash
# per-kernel averages of one bench run (no ViT / CPU legs):  bash tools/kstats_bench.sh [extra bench args]
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ksb && rocprofv3 --kernel-trace --stats -d /tmp/ksb -o x --output-format csv -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vit "$@" > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/ksb/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if any(k in n for k in ('selector', 'refine', 'sim_argmax', 'bn_tokens', 'preprocess', 'select_keypoints', 'intensity', 'finalize', 'decode')):
        print(f"{n[:72]:72s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:10.1f}")
PY
