#!/bin/bash
# Regenerates the profile artefacts of a round on the GPU box (run from the repo root):  bash tools/profile_round.sh r04 [a|b|c]
# (a: bench line, kernel stats, counters of the exact pass; b: the two ViTs; c: latency, yardsticks, sweeps, other workloads;
#  no letter: everything - more than one 20-minute gpurun call)
# Writes gpurun_out/<tag>_*: copy the ones to be judged into profiles/ afterwards.  Raw traces stay in /tmp on the box.
# rocprofv3 is always given the program itself after `--`; --pmc passes carry only --kernel-trace (no other trace domain).
set -e
TAG=${1:-r05}
PART=${2:-abc}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RX='selector_saliency|gather_refine|sim_argmax|bn_tokens|preprocess|select_keypoints|intensity_kernel|match_finalize|refine_bf16|selector_bf16|keys_decode'
VRX='gemm_rt_kernel|mlp_fused_kernel|attn_kernel|im2patch|prefix_rows|ln_rows'
BENCH="$ROOT/bench.py --steps 5 --warmup 2"
VIT="$ROOT/tools/bench_vit.py 448 82"

if [[ $PART == *a* ]]; then
echo "[1/7] bench line"; python $BENCH > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "[2/7] kernel stats of the same command, main leg only (every call of a kernel is then the same shape)"
MAIN="--no-cpu-baseline --no-vit --no-bf16 --no-upload --no-directory --sustain 0"
rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats -d /tmp/ks -o x --output-format csv -- python $BENCH $MAIN > $OUT/${TAG}_bench_profiled.json 2>/dev/null
cp $(find /tmp/ks -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
# warm-only: the 2 warm-up steps' launches dropped, so that flop / AverageNs / peak from this file == roofline.frac of the line
python $ROOT/tools/kstats_warm.py /tmp/ks $OUT/${TAG}_kernel_stats_warm.csv 2 > /dev/null
echo "[3/7] HBM traffic + stall counters (separate --pmc passes)"
rm -rf /tmp/pm1 /tmp/pm2 /tmp/pm3 /tmp/pm4
P="--kernel-trace --kernel-include-regex $RX --output-format csv -o x"
PB="--steps 2 --warmup 1 --no-cpu-baseline --no-vit --no-upload --no-directory --sustain 0"
rocprofv3 --pmc FETCH_SIZE $P -d /tmp/pm1 -- python $BENCH $PB > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE $P -d /tmp/pm2 -- python $BENCH $PB > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA $P -d /tmp/pm3 \
  -- python $BENCH $PB > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES $P -d /tmp/pm4 \
  -- python $BENCH $PB > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_summary.json /tmp/pm1 /tmp/pm2 /tmp/pm3 /tmp/pm4 > /dev/null
fi
if [[ $PART == *b* ]]; then
echo "[4/7] ViT alone: bench, kernel stats, counters of the HEAD kernels"
python $VIT > $OUT/${TAG}_vit_bench.txt
python $ROOT/tools/bench_vit.py 448 1 8 41 64 164 >> $OUT/${TAG}_vit_bench.txt
rm -rf /tmp/kv && rocprofv3 --kernel-trace --stats -d /tmp/kv -o x --output-format csv -- python $VIT > /dev/null 2>&1
cp $(find /tmp/kv -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_vit_kernel_stats.csv
rm -rf /tmp/pv1 /tmp/pv2 /tmp/pv3 /tmp/pv4
PV="--kernel-trace --kernel-include-regex $VRX --output-format csv -o x"
rocprofv3 --pmc FETCH_SIZE $PV -d /tmp/pv1 -- python $VIT > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE $PV -d /tmp/pv2 -- python $VIT > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU $PV -d /tmp/pv3 -- python $VIT > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE $PV -d /tmp/pv4 -- python $VIT > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_vit.json /tmp/pv1 /tmp/pv2 /tmp/pv3 /tmp/pv4 > /dev/null
echo "[4b] the fp32-operand ViT (reference numerics for A1, csrc/vit_f32.hip): bench, kernel stats, matrix-pipe counters"
export SSLAM_BENCH_VIT=fp32
python $ROOT/tools/bench_vit.py 448 8 41 83 166 > $OUT/${TAG}_vit_f32_bench.txt
rm -rf /tmp/kf && rocprofv3 --kernel-trace --stats -d /tmp/kf -o x --output-format csv -- python $ROOT/tools/bench_vit.py 448 83 > /dev/null 2>&1
cp $(find /tmp/kf -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_vit_f32_kernel_stats.csv
rm -rf /tmp/pf1 /tmp/pf2 /tmp/pf3
PF="--kernel-trace --kernel-include-regex gemm_f32|attn_f32_kernel|ln_rows_f32 --output-format csv -o x"
rocprofv3 --pmc FETCH_SIZE $PF -d /tmp/pf1 -- python $ROOT/tools/bench_vit.py 448 83 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE $PF -d /tmp/pf2 -- python $ROOT/tools/bench_vit.py 448 83 > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU $PF -d /tmp/pf3 -- python $ROOT/tools/bench_vit.py 448 83 > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_vit_f32.json /tmp/pf1 /tmp/pf2 /tmp/pf3 > /dev/null
unset SSLAM_BENCH_VIT
fi
if [[ $PART == *c* ]]; then
echo "[5/7] single-frame latency"
python $ROOT/tools/bench_latency.py > $OUT/${TAG}_latency.json 2>/dev/null
echo "[6/7] library yardstick + bandwidth probe (context for the roofline fractions, not product code)"
python $ROOT/tools/yardstick_vit.py > $OUT/${TAG}_yardstick.txt
python $ROOT/tools/bw_probe.py >> $OUT/${TAG}_yardstick.txt
echo "[7/7] round quantisation of the conv / descriptor MLP launches, two-stream ViT groups"
python $ROOT/tools/conv_rounds.py > $OUT/${TAG}_rounds.txt 2>/dev/null
python $ROOT/tools/refine_rounds.py >> $OUT/${TAG}_rounds.txt 2>/dev/null
python $ROOT/tools/vit_streams.py 82 8 >> $OUT/${TAG}_rounds.txt 2>/dev/null
echo "[8] round-3 additions: A2 at the three grids, host -> device feed (sweep + event timeline), other workloads"
python $ROOT/tools/bn_bench.py > $OUT/${TAG}_bn_grids.txt 2>/dev/null
python $ROOT/tools/m1_bench.py > $OUT/${TAG}_m1_shapes.txt 2>/dev/null
python $ROOT/tools/refine_bench.py > $OUT/${TAG}_refine_entries.txt 2>/dev/null
python $ROOT/tools/refine_bf16_bench.py >> $OUT/${TAG}_refine_entries.txt 2>/dev/null
python $ROOT/tools/gather_cost.py > $OUT/${TAG}_multi_rank_costs.txt 2>/dev/null
python $ROOT/tools/halo_cost.py >> $OUT/${TAG}_multi_rank_costs.txt 2>/dev/null
python $ROOT/tools/upload_sweep.py > $OUT/${TAG}_upload_sweep.txt 2>/dev/null
python $ROOT/tools/upload_timeline.py 307 > $OUT/${TAG}_upload_timeline.txt 2>/dev/null
echo "[9] round-4 additions: the drop-in backbone at the reference's batch sizes, the per-frame stepper, the 4-rank rehearsal"
python $ROOT/tools/backbone_batch_sweep.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_backbone_batch_sweep.txt
python $ROOT/tools/online_probe.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_online_stepper.txt
python $ROOT/bench.py --gpus 4 --rehearse-shared-gpu --steps 2 --warmup 1 > $OUT/${TAG}_rehearsal_4ranks.json 2>/dev/null
# (clock probes need probe builds: tools/build_variant.sh probe csrc/vit_f32.hip vit_f32.hip -DSSLAM_CLOCK_PROBE -> tools/vit_f32_probe.py;
#  tools/build_variant.sh rtprobe csrc/vit.hip vit.hip -DSSLAM_RT_PROBE -> SSLAM_BENCH_SO=... tools/rt_probe.py 82)
# every BASELINE configuration at its full per-GPU size (configs[2]: 2 965 frames; configs[3] / [4]: one GPU's share of the 4- / 8-GPU job)
for wl in fr1_xyz_50 fr2_desk_1024kp synthetic_2048kp fr3_long_office_4gpu synthetic_2048kp_8gpu; do
  python $ROOT/bench.py --steps 5 --warmup 2 --workload $wl --no-cpu-baseline --no-vit --no-directory --sustain 5 > $OUT/${TAG}_bench_$wl.json 2>/dev/null
done
fi
echo done
