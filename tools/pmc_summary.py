#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc / --kernel-trace CSV output directories into one small JSON (per kernel: calls, mean
duration, mean of every collected counter), so that the raw traces need not leave the GPU box.

usage: pmc_summary.py OUT.json DIR [DIR ...]
HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads, so hbm_read_bytes = 2 * FETCH_SIZE * 1024."""
import csv
import glob
import json
import os
import sys

KEEP = ("selector_saliency_halo_kernel", "selector_saliency", "gather_refine", "sim_argmax", "bn_tokens", "preprocess_kernel", "select_keypoints",
        "intensity_kernel", "match_finalize", "gather_kernel", "gemm_ares_kernel", "gemm_bf16_kernel", "attn_kernel", "ln_rows_kernel",
        "mlp_fused_kernel", "im2patch", "refine_bf16_kernel", "selector_bf16_kernel", "selector_bf16_halo_kernel", "preprocess_fast_kernel", "bn_tokens_reg_kernel", "keys_decode_kernel",
        "gemm_rt_kernel", "prefix_rows_kernel", "gemm_f32_rows_kernel", "gemm_f32_kernel", "attn_f32_kernel", "ln_rows_f32_kernel")


def short(name):
    for k in KEEP:
        if k in name:
            i = name.find(k)
            rest = name[i:].replace("(anonymous namespace)::", "")
            if rest[len(k):len(k) + 1] == "<":           # keep the template arguments: they tell the instantiations apart
                depth = 0
                for j, ch in enumerate(rest):
                    depth += ch == "<"
                    depth -= ch == ">"
                    if depth == 0 and ch == ">":
                        return rest[:j + 1]
            j = rest.find("(")
            return rest[:j if j > 0 else None]
    return None


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    res = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    e = res.setdefault(k, {}).setdefault("counters", {}).setdefault(r["Counter_Name"], [0.0, 0])
                    e[0] += float(r["Counter_Value"])
                    e[1] += 1
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    e = res.setdefault(k, {}).setdefault("dur_ns_" + os.path.basename(d.rstrip("/")), [0.0, 0])
                    e[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                    e[1] += 1
                    for fld in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size"):
                        if fld in r:
                            res[k][fld] = r[fld]
    for k, v in res.items():
        for name in list(v):
            if name.startswith("dur_ns_"):
                s, n = v[name]
                v[name] = {"mean": s / max(n, 1), "calls": n}
        if "counters" in v:
            v["counters"] = {c: s / max(n, 1) for c, (s, n) in v["counters"].items()}
            c = v["counters"]
            if "FETCH_SIZE" in c:
                v["hbm_read_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"] * 1024.0
            if "WRITE_SIZE" in c:
                v["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024.0
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v.get("counters") for k, v in res.items()}, indent=1)[:3000])


if __name__ == "__main__":
    main()
