#!/usr/bin/env python3
"""Prints the per-layer ViT table of DESIGN_HISTORY.md 5b from profiles/<tag>_vit_kernel_stats.csv, _pmc_vit.json and _yardstick.txt.
usage: vit_table.py [tag] [frames of the profiled launch group]"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 82
ks = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f"{ROOT}/profiles/{tag}_vit_kernel_stats.csv"))}
def pick(*keys):
    return next(v for k, v in ks.items() if all(x in k for x in keys))
qkv, attn, oproj, mlp = pick("gemm_rt_kernel<1, 3"), pick("attn_kernel"), pick("gemm_rt_kernel<1, 1", "ProBf16"), pick("mlp_fused")
lib = {}
for line in open(f"{ROOT}/profiles/{tag}_yardstick.txt"):
    m = re.match(r"(\w+)\s.*?:\s+([\d.]+) us", line)
    if m:
        lib[m.group(1)] = float(m.group(2))
sc = 64.0 / frames
print(f"frames {frames}: QKV {qkv:.0f} ({qkv*sc:.0f}/64fr, lib {lib['qkv']:.0f})  attn {attn:.0f} ({attn*sc:.0f}, lib {lib['sdpa']:.0f})  "
      f"o_proj {oproj:.0f} ({oproj*sc:.0f}, lib {lib['proj']:.0f})  MLP {mlp:.0f} ({mlp*sc:.0f}, lib {lib['fc1']:.0f}+{lib['fc2']:.0f})  "
      f"layer {qkv+attn+oproj+mlp:.0f} ({(qkv+attn+oproj+mlp)*sc:.0f}, lib {lib['qkv']+lib['sdpa']+lib['proj']+lib['fc1']+lib['fc2']:.0f})")
p = json.load(open(f"{ROOT}/profiles/{tag}_pmc_vit.json"))
for k, v in p.items():
    c = v.get("counters", {})
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
        d = [vv["mean"] for kk, vv in v.items() if kk.startswith("dur_ns_")]
        print(f"{k[:44]:44s} {sum(d)/len(d)/1e3:7.1f} us  mfma busy {c['SQ_VALU_MFMA_BUSY_CYCLES']/1024/1e3:6.0f} k of {c['GRBM_GUI_ACTIVE']/8/1e3:6.0f} k "
              f"({100*c['SQ_VALU_MFMA_BUSY_CYCLES']/1024/(c['GRBM_GUI_ACTIVE']/8):4.1f} %)  HBM in {v.get('hbm_read_bytes_per_launch',0)/1e6:6.0f} MB out {v.get('hbm_write_bytes_per_launch',0)/1e6:6.0f} MB")
