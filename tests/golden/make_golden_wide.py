#!/usr/bin/env python3
"""A wider, randomised golden set for the two order-sensitive stages, produced like make_golden.py by RUNNING the
reference's own code (KeypointSelector.select_keypoints and SequenceMatcher.match_with_quality) in the authoring
container.  Inputs are regenerated from seeds by the tests (tests/synth.py: wide_map, wide_pair); the fixture stores the
reference's outputs only.

  select_wide.npz : 64 tie-free saliency maps (uniform / band-around-0.5 / smooth), random grid, K, nms_radius, percentile
  match_wide.npz  : 32 descriptor pairs with duplicated rows (exact ties), random sizes and thresholds

Usage:  python tests/golden/make_golden_wide.py
"""
from __future__ import annotations

import numpy as np

import make_golden as mg
from make_golden import t  # noqa: F401
import synth  # (path set up by make_golden); the case generators live in tests/synth.py so that the tests rebuild the inputs


def main():
    M = mg._import_reference()
    sel = mg.load_selector(M["sel"], 0, 256)
    out = {}
    n_sel = 64
    for s in range(n_sel):
        m, K, radius, pct = synth.wide_map(s)
        try:
            kp, sc, idx = mg.run_select(sel, m, K, radius, pct)
            out[f"s{s}_idx"] = idx.astype(np.int16)
            out[f"s{s}_scores"] = sc.astype(np.float32)
        except RuntimeError:
            out[f"s{s}_idx"] = np.zeros(0, np.int16)          # the reference raises (topk k > cells)
            out[f"s{s}_scores"] = np.zeros(0, np.float32)
    out["count"] = n_sel
    mg.save("select_wide", **out)

    mq = M["vms"].SequenceMatcher.match_with_quality
    out = {}
    n_pair = 32
    for s in range(n_pair):
        d1, d2, s1, s2, kw = synth.wide_pair(s)
        mt, q = mq(d1, d2, s1, s2, **kw)
        out[f"p{s}_matches"] = mt.astype(np.int16)
        out[f"p{s}_quality"] = q.astype(np.float32)
    out["count"] = n_pair
    mg.save("match_wide", **out)


if __name__ == "__main__":
    main()
