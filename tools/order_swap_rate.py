#!/usr/bin/env python3
"""Keypoint-order agreement of the canonical fp32 evaluation (CPU oracle == HIP kernels, bit for bit) with the reference
(torch 2.10 CPU) on the end-to-end goldens at G = 40 / G = 60 and the 32-frame order set at G = 60.  Prints one JSON
document (committed as profiles/r04_order_swap_rate.json); the same comparisons are asserted by tests/test_oracle_golden.py
(CPU) and tests/test_gpu_parity.py (HIP).  CPU only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import e2e_check  # noqa: E402
import synth  # noqa: E402
from oracle import ora  # noqa: E402
from test_oracle_golden import _oracle_sequence  # noqa: E402


def main():
    out = {"near_tie_bar": e2e_check.NEAR_TIE}
    allg60 = []
    for tag in ("e2e_g40", "e2e_g60"):
        rep, _ = _oracle_sequence(tag)
        out[tag] = dict(summary=e2e_check.summarise(rep["frames"]), frames=rep["frames"],
                        pairs=dict(count=len(rep["pairs"]), index_exact=sum(p["index_exact"] for p in rep["pairs"]),
                                   cells_equal=sum(p["cells_equal"] for p in rep["pairs"]),
                                   matches=sum(p["matches"] for p in rep["pairs"])))
        if tag == "e2e_g60":
            allg60 += rep["frames"]

    def idx_of(i, seed):
        feat = ora.bn_tokens(synth.tokens(seed, 60))[0].reshape(1, 60, 60, 384)
        return ora.select_keypoints(ora.selector_saliency(feat, synth.selector_state(0)), 2048)[2][0]
    fr = e2e_check.check_order_set(idx_of)
    out["order_g60"] = dict(summary=e2e_check.summarise(fr), frames=fr)
    allg60 += fr
    out["g60_all_48_frames"] = e2e_check.summarise(allg60)
    out["g60_all_48_frames"]["swapped_fraction_of_positions"] = out["g60_all_48_frames"]["swapped_positions_mean"] / 2048
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
