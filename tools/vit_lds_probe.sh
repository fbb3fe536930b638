#!/bin/bash
# bf16 HIP ViT alone at 82 frames: frames/s, per-kernel table, LDS bank-conflict counters:  bash tools/vit_lds_probe.sh TAG
TAG=${1:-vitlds}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
VIT="$ROOT/tools/bench_vit.py 448 82"
VRX='gemm_rt_kernel|mlp_fused_kernel|attn_kernel|im2patch|prefix_rows|ln_rows'
python $VIT > $OUT/${TAG}_vit_bench.txt 2>&1
rm -rf /tmp/kv && rocprofv3 --kernel-trace --stats -d /tmp/kv -o x --output-format csv -- python $VIT > /dev/null 2>&1
cp $(find /tmp/kv -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_vit_kernel_stats.csv
rm -rf /tmp/pv4
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE \
  --kernel-trace --kernel-include-regex "$VRX" --output-format csv -o x -d /tmp/pv4 -- python $VIT > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_vit_lds.json /tmp/pv4 > /dev/null
python - <<PY
import json
d = json.load(open("$OUT/${TAG}_pmc_vit_lds.json"))
for k, v in d.items():
    c, a = v.get("SQ_LDS_BANK_CONFLICT"), v.get("SQ_LDS_IDX_ACTIVE")
    if c is not None and a:
        print(f"{k[:90]:90s} conflict/active = {c / a:.3f}")
PY
cat $OUT/${TAG}_vit_bench.txt
python $ROOT/tools/kstats.py /tmp/kv 8
